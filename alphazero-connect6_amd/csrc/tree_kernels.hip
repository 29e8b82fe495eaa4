// tree_kernels.hip -- K2: batched AlphaZero MCTS (select / expand / backup / move / prune) and
// K3: dihedral augment / recover, one wave (64 lanes) per game.
//
// Reference being reproduced (Connect6/libconnect6/src/policy/alphazero_policy/mod.rs):
//   maximum_from :243-296   select :313-323   expand :331-381   update :386-408
//   search :411-426         policy :433-452   next (prune) :464-489   get_from :218-234
// and augment.rs:60-92, agent/agent_impl/mod.rs:121-178 (game loop).
//
// Data structure.  The reference keeps a HashMap<board hash, Node> in which EVERY legal child of
// every expanded node is a full Node.  Only three facts about an unexpanded node are ever read:
// visit == 0, q_value == 0 and n_prob (the prior written by the LAST expanded parent, :366-370).
// So the engine stores records only for EXPANDED nodes, each carrying prior[cell] = n_prob of its
// child at that cell, plus a table  Zobrist key -> record  through which a child's shared
// visit / q_value are found (children reached by different move orders are one node, as in the
// reference).  The last-writer rule is kept exact: when a node Q is expanded, every expanded node
// P with P + i == Q + s (one stone swapped) gets prior_P[i] = prob_Q[s].  P = Q - i + s holds every
// stone of the root only if i is one of the stones the select path added to the root -- a P without
// a root stone can never be reached again, its priors are never read -- so the candidates are
// (path stones of the mover's colour) x (empty cells of Q), each looked up by key.  Round 1 scanned
// all records of the game for every expansion (26 bytes per record and two 64-bit divisions).
//
// Storage is paged (c6_engine.h): a game's records sit in pages of 64 from a pool shared by all games, its table in chunks of 1024
// slots from a second pool.  The wave copies its game's page table and chunk list (a few dozen words) into LDS with the tag row, so a
// record / table-slot address costs one LDS read.  Pages and chunks change hands at MOVE boundaries only (post_move / reset:
// reprovision_wave): a game enters a move with >= S free records and a table of >= 2 x (live + S) slots, so no search ever waits.
//
// Memory accesses are what a search costs (profiles/r02_tree_traffic.txt), so:
//  * every table slot has a TAG byte (c6_ht_tag_of); the wave loads the game's whole tag row into LDS
//    once per search (H bytes, contiguous): looking up an absent child -- most children of most
//    nodes -- touches no memory, a present one costs its one 16-byte entry, an insert finds its
//    slot without a probe;
//  * the root's children, scanned by every search, are kept as one contiguous row (C6RootChild).
//
// A lane owns cells lane, lane+64, ... (row-major order == child order of the reference, which
// lists children in `possible` order, simulate/mod.rs:58-69).
//
// Launches of a round.  Built-in evaluators: c6_search_kernel (last round's move finished, select, evaluate, expand + backup,
// policy() -- one launch, the leaf stays in registers) + K1.  Net / callback evaluators: c6_select_kernel (head + select) ->
// evaluator -> c6_expand_kernel (expand + backup + policy()) + K1.  The move K1 applies is finished (root advanced, pruned,
// game ended / recycled) at the head of the next round's first kernel, or by c6_post_move_kernel at the end of a step call.
#include "c6_device.h"
#include "c6_engine.h"

// Diagnostic builds only (tools/ab_build.sh stamps tree_kernels.hip -DC6_AB_STAMPS; tools/tree_stamps.py): where a wave's cycles go.  C6_STAMP(a, b) ends
// phase a and begins phase b: lane 0 adds s_memtime to one accumulator and subtracts it from the other, so acc[p] = sum over waves of the cycles
// (waits included) spent in p.  No stamp exists in the product.
#ifdef C6_AB_STAMPS
__device__ unsigned long long c6_stamp_acc[16];
__shared__ unsigned long long c6_stamp_lds[16];      // one wave per workgroup (WAVES_PER_BLOCK 1): the wave's own accumulators, flushed once
#define C6_STAMP(a, b)                                                                    \
    do {                                                                                  \
        if (lane == 0) {                                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                   \
            c6_stamp_lds[a] += t_;                                                        \
            c6_stamp_lds[b] -= t_;                                                        \
        }                                                                                 \
    } while (0)
#define C6_STAMP_INIT() do { if (lane < 16) c6_stamp_lds[lane] = 0ull; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); } while (0)
#define C6_STAMP_FLUSH() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); if (lane < 16) atomicAdd(&c6_stamp_acc[lane], c6_stamp_lds[lane]); } while (0)
extern "C" int c6_debug_stamps(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(c6_stamp_acc), 16 * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(c6_stamp_acc), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#else
#define C6_STAMP(a, b) do { } while (0)
#define C6_STAMP_INIT() do { } while (0)
#define C6_STAMP_FLUSH() do { } while (0)
#endif
enum { C6_PH_HEAD = 0, C6_PH_SCAN, C6_PH_NOISE, C6_PH_SCORE, C6_PH_STEP, C6_PH_EVAL, C6_PH_EXPAND, C6_PH_MOVE, C6_PH_TAIL, C6_PH_OUT };

// One wave = one game = one workgroup (r4).  With four waves per workgroup a CU's slot came free only when the slowest of four games had
// finished its rounds: config 5 ran 0.1507 ms per round; two waves 0.143; one wave 0.139 (same box, tools/ab_many.sh).
#ifndef WAVES_PER_BLOCK
#define WAVES_PER_BLOCK 1
#endif
#ifndef C6_OCC2
#define C6_OCC2 3      // c6_search_kernel<2, .>: at 4 blocks per SIMD it spills 19 registers and is 17 % slower (tools/ab_build.sh -DC6_OCC2=4)
#endif
#ifndef C6_OCC4
#define C6_OCC4 2
#endif

// ---- the table: board key -> record, with the child statistics select reads riding in the entry ---------------------------
// C6HtEntry (c6_engine.h) = {key, idx + 1, visit (saturating at 65535), q}: a child's visit count and q_value arrive with the
// probe that finds it -- one memory round trip per level instead of probe + gather.  The authoritative copies stay in
// rec_stat; expand and backup update both (rec_stat[record].slot = its table slot).

__device__ __forceinline__ uint32_t ht_home(unsigned long long key, int H) { return (uint32_t)(key ^ (key >> 32)) & (uint32_t)(H - 1); }

// one 16-byte load of an entry
__device__ __forceinline__ C6HtEntry ht_load(const C6HtEntry *ep) {
    const uint4 v = *reinterpret_cast<const uint4 *>(ep);
    C6HtEntry e;
    e.key = ((unsigned long long)v.y << 32) | v.x;
    e.idx1 = (uint16_t)(v.z & 0xffffu);
    e.visit = (uint16_t)(v.z >> 16);
    e.q = __uint_as_float(v.w);
    return e;
}
__device__ __forceinline__ void ht_store(C6HtEntry *ep, unsigned long long key, int idx, int visit, float q) {
    const uint32_t v = (uint32_t)(visit > 65535 ? 65535 : visit);
    *reinterpret_cast<uint4 *>(ep) = uint4{(uint32_t)key, (uint32_t)(key >> 32), (uint32_t)(idx + 1) | (v << 16), __float_as_uint(q)};
}

__device__ __forceinline__ uint16_t sat16(int v) { return (uint16_t)(v > 65535 ? 65535 : v); }

// The wave's view of its game's paged storage: chunk list and page table in LDS, and the tag row too when the game's table has at
// most C6_TAG_LDS_MAX slots (else the tags are read in memory, through the chunk list).  Tags mirror the table exactly: tag 0 <=>
// empty slot.
typedef __attribute__((address_space(3))) uint8_t c6_lds_u8;
typedef __attribute__((address_space(3))) uint32_t c6_lds_u32;
struct C6Tags {
    c6_lds_u8 *l;          // tag row (lds)
    c6_lds_u32 *ck, *pt;   // logical chunk -> pool chunk, logical page -> pool page
    uint8_t *tagp;         // E.ht_tag
    C6HtEntry *htp;        // E.ht
    bool lds;
    uint32_t mask;         // hslots - 1
    __device__ __forceinline__ size_t slot_at(uint32_t s) const { return ((size_t)ck[s >> C6_CHUNK_SHIFT] << C6_CHUNK_SHIFT) | (size_t)(s & (C6_CHUNK_SLOTS - 1)); }
    __device__ __forceinline__ C6HtEntry *entry(uint32_t s) const { return htp + slot_at(s); }
    __device__ __forceinline__ size_t rec(int r) const { return ((size_t)pt[r >> C6_PAGE_SHIFT] << C6_PAGE_SHIFT) | (size_t)(r & (C6_PAGE_RECS - 1)); }
    __device__ __forceinline__ uint8_t at(uint32_t s) const {
        if (lds) return l[s];
        return tagp[slot_at(s)];
    }
    __device__ __forceinline__ void set(uint32_t s, uint8_t t) const {
        if (lds) l[s] = t;
        tagp[slot_at(s)] = t;
    }
};
// LDS of one wave: [tag row: min(hcap, C6_TAG_LDS_MAX) bytes][chunk list: hcap / 1024 words][page table: pages_cap words]
__host__ __device__ __forceinline__ int c6_tagrow_bytes(int hcap) { return hcap < C6_TAG_LDS_MAX ? hcap : C6_TAG_LDS_MAX; }
__host__ __device__ __forceinline__ int c6_cklist_bytes(int hcap) { return (((hcap >> C6_CHUNK_SHIFT) * 4) + 15) & ~15; }
__host__ __device__ __forceinline__ int c6_wave_lds(int hcap, int pages_cap) { return c6_tagrow_bytes(hcap) + c6_cklist_bytes(hcap) + ((pages_cap * 4 + 15) & ~15); }
// bind T to the wave's LDS area without loading anything (the rare paths work on the global tables)
__device__ __forceinline__ void tags_bind(const C6Dev &E, uint8_t *lds, C6Tags &T) {
    T.l = (c6_lds_u8 *)lds;
    T.ck = (c6_lds_u32 *)(lds + c6_tagrow_bytes(E.hcap));
    T.pt = (c6_lds_u32 *)(lds + c6_tagrow_bytes(E.hcap) + c6_cklist_bytes(E.hcap));
    T.tagp = E.ht_tag;
    T.htp = E.ht;
}
// the game's chunk list, page table and (small tables) tag row into the wave's LDS
__device__ __forceinline__ void tags_load(const C6Dev &E, int g, int lane, uint8_t *lds, C6Tags &T) {
    tags_bind(E, lds, T);
    const int H = (int)E.hdr[g].hslots, np = (int)E.hdr[g].npages;
    T.mask = (uint32_t)(H - 1);
    T.lds = H <= C6_TAG_LDS_MAX;
    const int nck = H >> C6_CHUNK_SHIFT;
    const uint32_t *ckg = E.chunk_table + (size_t)g * E.maxchunks, *ptg = E.page_table + (size_t)g * E.maxpages;
    for (int i = lane; i < nck; i += 64) T.ck[i] = ckg[i];
    for (int i = lane; i < np; i += 64) T.pt[i] = ptg[i];
    if (T.lds) {
        typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
        typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
        lds_u32x4 *dst = (lds_u32x4 *)lds;
        for (int c = 0; c < nck; c++) {      // a chunk's 1024 tags: one 16-byte load per lane
            const u32x4 *src = reinterpret_cast<const u32x4 *>(E.ht_tag + ((size_t)ckg[c] << C6_CHUNK_SHIFT));
            dst[c * (C6_CHUNK_SLOTS / 16) + lane] = src[lane];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
// from `slot` on: the first slot whose tag is t (true) or that is empty (false); `budget` bounds the walk by the table size
__device__ __forceinline__ bool tag_seek(const C6Tags &T, uint8_t t, uint32_t &slot, int &budget) {
    while (budget-- > 0) {
        const uint8_t x = T.at(slot);
        if (x == t) return true;
        if (x == 0) return false;
        slot = (slot + 1) & T.mask;
    }
    return false;
}
// continue a lookup whose first tag match `slot` has been loaded into `e`: true and the entry when `key` is in the table
__device__ __forceinline__ bool ht_verify(const C6Tags &T, unsigned long long key, uint8_t t, uint32_t &slot, int &budget, C6HtEntry &e) {
    while (e.key != key) {      // a different key with the same tag (1 in 255): walk on
        slot = (slot + 1) & T.mask;
        if (!tag_seek(T, t, slot, budget)) return false;
        e = ht_load(T.entry(slot));
    }
    return true;
}
__device__ __forceinline__ bool ht_find(const C6Tags &T, unsigned long long key, C6HtEntry &e) {
    const uint8_t t = c6_ht_tag_of(key);
    uint32_t slot = ht_home(key, (int)T.mask + 1);
    int budget = (int)T.mask + 1;
    if (!tag_seek(T, t, slot, budget)) return false;
    e = ht_load(T.entry(slot));
    return ht_verify(T, key, t, slot, budget, e);
}
// insert by the game's one writer (the tags say where the first empty slot is: no probe); returns the slot, -1 = table full
__device__ __forceinline__ int ht_insert(const C6Tags &T, unsigned long long key, int idx, int visit, float q) {
    uint32_t slot = ht_home(key, (int)T.mask + 1);
    for (int it = 0; it <= (int)T.mask; it++) {
        if (T.at(slot) == 0) {
            ht_store(T.entry(slot), key, idx, visit, q);
            T.set(slot, c6_ht_tag_of(key));
            return (int)slot;
        }
        slot = (slot + 1) & T.mask;
    }
    return -1;
}
// concurrent insert by the lanes of the game's wave (distinct keys) into a table being rebuilt: memory only (global chunk list `ckg`)
__device__ __forceinline__ int ht_insert_atomic(C6HtEntry *htp, uint8_t *tagp, const uint32_t *ckg, int H, unsigned long long key, int idx, int visit, float q) {
    uint32_t slot = ht_home(key, H);
    for (int it = 0; it < H; it++) {
        const size_t at = ((size_t)ckg[slot >> C6_CHUNK_SHIFT] << C6_CHUNK_SHIFT) | (size_t)(slot & (C6_CHUNK_SLOTS - 1));
        unsigned long long prev = atomicCAS(&htp[at].key, 0ull, key);
        if (prev == 0ull || prev == key) {
            htp[at].idx1 = (uint16_t)(idx + 1);
            htp[at].visit = sat16(visit);
            htp[at].q = q;
            tagp[at] = c6_ht_tag_of(key);
            return (int)slot;
        }
        slot = (slot + 1) & (uint32_t)(H - 1);
    }
    return -1;
}

__device__ __forceinline__ void next_turn(int &turn, int &remain) {   // simulate/mod.rs:268-273
    remain -= 1;
    if (remain <= 0) { remain = 2; turn = -turn; }
}

// what select hands to the evaluator / expand: in registers inside the fused search kernel, through the leaf_* arrays otherwise
template <int J>
struct C6Leaf {
    int kind, turn, level, len;
    unsigned long long key;
    int my_path;            // lane i < 64: record of the path's i-th step | the cell chosen there << 16
    int8_t st[J];           // the leaf position, lane's cells
};

// Everything a search() needs that is addressed by the game alone, fetched as ONE batch of independent loads (the kernels are chains of
// dependent memory round trips; every scalar loaded where it is first used would be a trip of its own).
template <int J>
struct C6GameRegs {
    unsigned long long key;
    int root_rec, turn, remain, level, top, sim_in_move, rootc_ok;
    uint32_t uid, nsearch;
    int alloc_page;         // logical page the next expansion takes its record from
    unsigned long long alloc_mask;      // that page's free mask (page_free), fetched behind the page table: free_record_prefetch
    int8_t st[J];           // the root position, lane's cells
};
template <int J>
__device__ __forceinline__ void load_game_regs(const C6Dev &E, int g, int lane, C6GameRegs<J> &R) {
    R.key = E.hdr[g].root_key;
    R.root_rec = E.hdr[g].root_rec;
    R.rootc_ok = (int)E.hdr[g].rootc_ok;
    R.turn = E.turn[g]; R.remain = E.remain[g]; R.level = E.hdr[g].nstones;
    R.uid = E.hdr[g].game_uid; R.nsearch = E.hdr[g].nsearch;
    R.top = E.hdr[g].free_top; R.sim_in_move = E.hdr[g].sim_in_move;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c = lane + 64 * j;
        R.st[j] = c < E.cells ? E.root_board[(size_t)g * E.cellsP + c] : (int8_t)2;
    }
    // wave-uniform values belong in scalar registers (the loads above are vector loads of one address)
#define C6_SGPR(x) x = __builtin_amdgcn_readfirstlane(x)
    R.key = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(R.key >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)R.key);
    C6_SGPR(R.root_rec); C6_SGPR(R.turn); C6_SGPR(R.remain); C6_SGPR(R.level); C6_SGPR(R.top); C6_SGPR(R.sim_in_move); C6_SGPR(R.rootc_ok);
    R.uid = __builtin_amdgcn_readfirstlane(R.uid); R.nsearch = __builtin_amdgcn_readfirstlane(R.nsearch);
    R.alloc_page = (int)E.hdr[g].alloc_page;
    C6_SGPR(R.alloc_page);
    R.alloc_mask = 0ull;
#undef C6_SGPR
}
// the free mask of the page the next expansion allocates from (second trip, overlapped with the first select level); T.pt is loaded
template <int J>
__device__ __forceinline__ void free_record_prefetch(const C6Dev &E, const C6Tags &T, C6GameRegs<J> &R) {
    unsigned long long m = R.top > 0 ? E.page_free[T.pt[R.alloc_page]] : 0ull;
    R.alloc_mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
}

// colour of the stone the select path places at step m, from the root's (turn, remain): simulate/mod.rs:268-273 unrolled
__device__ __forceinline__ int path_colour(int root_turn, int root_remain, int m) {
    if (m < root_remain) return root_turn;
    return (((m - root_remain) >> 1) & 1) ? root_turn : -root_turn;
}

// ------------------------------------------------------------------------------------------
// select: walk from the root by PUCT + Dirichlet noise until a node without record (leaf)
// ------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ void select_wave(const C6Dev &E, int g, int lane, double *gscratch, const C6Tags &T, const C6GameRegs<J> &R, C6Leaf<J> &L) {
    const int cells = E.cells;
    const int H = (int)T.mask + 1;

    unsigned long long key = R.key;
    int p = R.root_rec;
    int turn = R.turn, remain = R.remain, level = R.level;
    const uint32_t k0 = E.seed_lo, k1 = E.seed_hi + R.uid, t = R.nsearch;

#pragma unroll
    for (int j = 0; j < J; j++) L.st[j] = R.st[j];

    int d = 0;
    int kind = C6_LEAF_EVAL;
    L.my_path = 0;
    while (p >= 0) {
        C6_STAMP(C6_PH_STEP, C6_PH_SCAN);
        const int nchild = cells - level;
        if (nchild == 0) break;                 // cannot happen: full boards are never records
        if (d >= E.maxd) { kind = -1; break; }  // cannot happen: a path adds a stone per step and maxd >= cells
        const unsigned long long *zk = E.zob + (turn == C6_WHITE ? C6_MAXCELLS : 0);
        const float *prior = E.rec_prior + T.rec(p) * (size_t)cells;

        int cidx[J], cvis[J];
        float cq[J], cpr[J];
        unsigned long long zmine[J];      // the Zobrist keys of the lane's cells: the chosen one's is read from its owner's register below (r4: `zk[chosen]`
        int vs = 0;                       // after the choice was a memory round trip of its own on the way to the next level)
        if (d == 0 && R.rootc_ok) {
            // the root: its children are one contiguous row
            const uint2 *rc = reinterpret_cast<const uint2 *>(E.rootc + (size_t)g * E.cellsP);
            uint2 v[J];
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c = lane + 64 * j;
                const bool mine = L.st[j] == 0;
                v[j] = rc[mine ? c : 0];
                cpr[j] = mine ? prior[c] : 0.f;
                zmine[j] = zk[mine ? c : 0];
            }
#pragma unroll
            for (int j = 0; j < J; j++) {
                cidx[j] = -1; cvis[j] = 0; cq[j] = 0.f;
                if (L.st[j] == 0 && (v[j].x & 0xffffu) != 0u) {
                    cidx[j] = (int)(v[j].x & 0xffffu) - 1;
                    cvis[j] = (v[j].x >> 16) == 65535u ? E.rec_stat[T.rec(cidx[j])].visit : (int)(v[j].x >> 16);
                    cq[j] = __uint_as_float(v[j].y);
                    vs += cvis[j];
                }
            }
        } else {
            // tags first (LDS): only children that may be present go to memory, all of the level's loads leave together
            unsigned long long ck[J];
            uint32_t cslot[J];
            int budget[J];
            bool hit[J];
            C6HtEntry ce[J];
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c = lane + 64 * j;
                const bool mine = L.st[j] == 0;
                zmine[j] = zk[mine ? c : 0];
                ck[j] = key ^ zmine[j];
                cslot[j] = ht_home(ck[j], H);
                budget[j] = H;
                hit[j] = mine && tag_seek(T, c6_ht_tag_of(ck[j]), cslot[j], budget[j]);
            }
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c = lane + 64 * j;
                if (hit[j]) ce[j] = ht_load(T.entry(cslot[j]));
                cpr[j] = L.st[j] == 0 ? prior[c] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < J; j++) {
                cidx[j] = -1; cvis[j] = 0; cq[j] = 0.f;
                if (hit[j] && ht_verify(T, ck[j], c6_ht_tag_of(ck[j]), cslot[j], budget[j], ce[j])) {
                    cidx[j] = (int)ce[j].idx1 - 1;
                    cvis[j] = ce[j].visit == 65535 ? E.rec_stat[T.rec(cidx[j])].visit : (int)ce[j].visit;
                    cq[j] = ce[j].q;
                    vs += cvis[j];
                }
            }
            if (d == 0) {      // first scan of this root: keep its children as a row for the searches to come
                C6RootChild *rc = E.rootc + (size_t)g * E.cellsP;
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c = lane + 64 * j;
                    if (c < cells)
                        *reinterpret_cast<uint2 *>(rc + c) = uint2{(uint32_t)(cidx[j] + 1) | ((uint32_t)sat16(cvis[j]) << 16), __float_as_uint(cq[j])};
                }
                if (lane == 0) E.hdr[g].rootc_ok = 1u;
            }
        }

        int chosen;
        if (nchild == 1) {                       // heuristic, mod.rs:255-258
            int mine = -1;
#pragma unroll
            for (int j = 0; j < J; j++) if (L.st[j] == 0) mine = lane + 64 * j;
            chosen = c6_wave_max_i32(mine);
        } else {
            const float visit_sum = (float)c6_wave_sum_i32(vs);
            // Dirichlet noise: Gamma draws normalised by their sum (C6RNG v4), mod.rs:260-262,283
            c6_gam gam[J];
            float noise_[J];
            bool need[J];
#pragma unroll
            for (int j = 0; j < J; j++) need[j] = L.st[j] == 0;
#ifdef C6_AB_NOISE_TWICE   // ablation builds only: the noise of another counter first, kept alive -- the difference to the tree is one noise evaluation
            c6_wave_gamma<J>(need, gam, gscratch, lane, E.alpha, E.gam_d, E.gam_c, k0, k1, t, (uint32_t)d + 128u);
            c6_wave_dirichlet<J>(gam, noise_);
#pragma unroll
            for (int j = 0; j < J; j++) asm volatile("" ::"v"(noise_[j]));
#endif
            C6_STAMP(C6_PH_SCAN, C6_PH_NOISE);
            c6_wave_gamma<J>(need, gam, gscratch, lane, E.alpha, E.gam_d, E.gam_c, k0, k1, t, (uint32_t)d);
            c6_wave_dirichlet<J>(gam, noise_);
            C6_STAMP(C6_PH_NOISE, C6_PH_SCORE);
            float sc[J], best = -INFINITY;
#pragma unroll
            for (int j = 0; j < J; j++) {
                sc[j] = -INFINITY;
                if (L.st[j] == 0) {                // mod.rs:265-280
                    const float noise = noise_[j];
                    const float visit = (float)cvis[j];
                    const float prob = E.eps * noise + (1.f - E.eps) * cpr[j];
                    const float puct = prob * sqrtf(visit_sum - visit) / (1.f + visit);
                    const float q = turn == C6_BLACK ? -cq[j] : cq[j];
                    sc[j] = q + E.cpuct * puct;
                    best = sc[j] > best ? sc[j] : best;
                }
            }
            best = c6_wave_max_f32(best);
            // exact ties -> uniform choice (mod.rs:287-295), C6RNG TIE draw
            unsigned long long tie[J];
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < J; j++) {
                tie[j] = __ballot(L.st[j] == 0 && sc[j] == best);
                cnt += __popcll(tie[j]);
            }
            int pick = 0;
            if (cnt > 1) {
                const c6_u4 w = c6_philox(0u, 0u, t, C6_TAG_TIE | ((uint32_t)d << 8), k0, k1);
                pick = (int)(c6_open01(w.x, w.y) * (double)cnt);
                if (pick >= cnt) pick = cnt - 1;
            }
            chosen = -1;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int pc = __popcll(tie[j]);
                if (chosen < 0) {
                    if (pick < pc) chosen = c6_nth_set_bit(tie[j], pick) + 64 * j;
                    else pick -= pc;
                }
            }
            if (chosen < 0) { kind = -1; break; }   // NaN scores: the reference panics (partial_cmp().unwrap())
            C6_STAMP(C6_PH_SCORE, C6_PH_SCAN);
        }
        C6_STAMP(C6_PH_SCAN, C6_PH_STEP);

        // step into the child: diff_board + simulate_in (mod.rs:417-420)
        const int cl = chosen & 63, cj = chosen >> 6;
        int child = -1;
#pragma unroll
        for (int j = 0; j < J; j++) if (j == cj) child = cidx[j];
        child = __builtin_amdgcn_readlane(child, __builtin_amdgcn_readfirstlane(cl));      // `chosen` is wave-uniform: a register read, not an LDS permute
        const int step = p | (chosen << 16);
        if (d < 64) { if (lane == d) L.my_path = step; }
        else if (lane == 0) E.path_rec[(size_t)g * E.maxd + d] = (uint32_t)step;      // beyond 64 steps: through memory
        d++;
        {
            unsigned long long zc = 0ull;
#pragma unroll
            for (int j = 0; j < J; j++) if (j == cj) zc = zmine[j];
            const int cls_ = __builtin_amdgcn_readfirstlane(cl);
            key ^= ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(zc >> 32), cls_) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)zc, cls_);
        }
#pragma unroll
        for (int j = 0; j < J; j++) if (j == cj && lane == cl) L.st[j] = (int8_t)turn;
        next_turn(turn, remain);
        level++;
        p = child;
    }
    if (kind == C6_LEAF_EVAL && level == cells) kind = C6_LEAF_TERMINAL;   // board full, mod.rs:338
    L.kind = kind; L.turn = turn; L.level = level; L.len = d; L.key = key;
}

// the game stops (NaN scores: the reference panics) -- reported through status, never silently
__device__ __forceinline__ void select_failed(const C6Dev &E, int g, int lane, int kind) {
    if (lane == 0) {
        E.hdr[g].status = C6_ST_ERROR;
        E.hdr[g].leaf_kind = C6_LEAF_NONE;
    }
}

// ------------------------------------------------------------------------------------------
// recover_way8 + mask (get_from, mod.rs:218-234) -- shared by every evaluator kind
// ------------------------------------------------------------------------------------------
// p(k, cell) returns image k's policy at `cell`; v(k) image k's value.
template <typename PF, typename VF>
__device__ __forceinline__ void recover_wave(int cells, int lane, const int16_t *rec_map, const int8_t *board, PF p, VF v,
                                             float *out_value, float *out_prior) {
    for (int c = lane; c < cells; c += 64) {
        float total = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {   // augment.rs:79-88: (p[2i] + flipV(p[2i+1])) rotated right i+1 times
            const float a = p(2 * i, (int)rec_map[(2 * i) * cells + c]);
            const float b = p(2 * i + 1, (int)rec_map[(2 * i + 1) * cells + c]);
            total = total + (a + b);
        }
        total = total / 8.f;
        out_prior[c] = total * (float)(board[c] == 0);   // masking, mod.rs:226-231
    }
    if (lane == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) s = s + v(k);        // iter().sum::<f32>() / 8., mod.rs:224
        *out_value = s / 8.f;
    }
}

// RandomEvaluator (mod.rs:137-169) / uniform evaluator, evaluated on device: get_from's tail (recover_way8 + mask) of the lane's
// own cells, in registers.  Same arithmetic as recover_wave (the summation order of augment.rs:79-88).
template <int J>
__device__ __forceinline__ void eval_builtin_regs(const C6Dev &E, int g, int lane, int kind, uint32_t t, int leaf_turn, const int8_t (&st)[J], float (&pr)[J],
                                                  float &value) {
    const int cells = E.cells;
    const uint32_t k0 = E.seed_lo, k1 = E.seed_hi + E.hdr[g].game_uid;
    if (kind == 3 && E.hdr[g].game_uid % (uint32_t)E.stress_every == 0u) {
        // C6_EVAL_STRESS (connect6_amd.h): prior 2^(-20 min(d2 - d2min, 6)) on the empty cells, d2 = squared distance to the board centre,
        // value -0.875 * side to move.  The function is invariant under the 8 dihedral images and its values are powers of two, so the
        // average of recover_way8 (augment.rs:79-88) returns them exactly.
        const int N = E.N, ctr = N / 2;
        int d2[J], d2min = 1 << 20;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j, r = c / N, q = c - r * N;
            d2[j] = (r - ctr) * (r - ctr) + (q - ctr) * (q - ctr);
            if (c < cells && st[j] == 0 && d2[j] < d2min) d2min = d2[j];
        }
        d2min = c6_wave_min_i32(d2min);
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j;
            int k = d2[j] - d2min;
            k = k > 6 ? 6 : k;
            const float p = __uint_as_float((uint32_t)(127 - 20 * k) << 23);      // 2^(-20 k), a normal float
            float total = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) total = total + (p + p);
            total = total / 8.f;
            pr[j] = c < cells && st[j] == 0 ? total : 0.f;
        }
        const float v = -0.875f * (float)leaf_turn;
        float sv = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) sv = sv + v;
        value = sv / 8.f;
        return;
    }
    if (kind == 1) {
        auto pf = [&](int k, int cell) -> float {
            const c6_u4 w = c6_philox((uint32_t)(cell >> 2), (uint32_t)k, t, C6_TAG_EVAL_P, k0, k1);
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
            return c6_u11(ws[cell & 3]);
        };
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j;
            pr[j] = 0.f;
            if (c < cells) {
                float total = 0.f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float a = pf(2 * i, (int)E.rec_map[(2 * i) * cells + c]);
                    const float b = pf(2 * i + 1, (int)E.rec_map[(2 * i + 1) * cells + c]);
                    total = total + (a + b);
                }
                total = total / 8.f;
                pr[j] = total * (float)(st[j] == 0);
            }
        }
        float sv = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) sv = sv + c6_u11(c6_philox((uint32_t)k, 0u, t, C6_TAG_EVAL_V, k0, k1).x);
        value = sv / 8.f;
    } else {
        const float u = 1.0f / (float)cells;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j;
            float total = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) total = total + (u + u);
            total = total / 8.f;
            pr[j] = c < cells ? total * (float)(st[j] == 0) : 0.f;
        }
        value = 0.f;
    }
}

// net / callback evaluators: img_pol [G][8][cells], img_val [G][8]
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void c6_recover_kernel(C6Dev E) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (g >= E.G || E.hdr[g].leaf_kind != C6_LEAF_EVAL) return;
    const int cells = E.cells;
    const float *pol = E.img_pol + (size_t)g * 8 * cells;
    const float *val = E.img_val + (size_t)g * 8;
    auto pf = [&](int k, int cell) -> float { return pol[k * cells + cell]; };
    auto vf = [&](int k) -> float { return val[k]; };
    recover_wave(cells, lane, E.rec_map, E.leaf_board + (size_t)g * E.cellsP, pf, vf, &E.hdr[g].eval_value,
                 E.eval_prior + (size_t)g * cells);
}

// augment_way8 (augment.rs:60-73) of the leaf boards as f32 for the host callback
// (board_to_float, cppbind/.../cpp_support.rs:10-18): img[g][k][c]
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void c6_augment_kernel(C6Dev E, float *img) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (g >= E.G || E.hdr[g].leaf_kind != C6_LEAF_EVAL) return;
    const int cells = E.cells;
    const int8_t *board = E.leaf_board + (size_t)g * E.cellsP;
    for (int k = 0; k < 8; k++)
        for (int c = lane; c < cells; c += 64) img[((size_t)g * 8 + k) * cells + c] = (float)board[E.aug_map[k * cells + c]];
}

// ------------------------------------------------------------------------------------------
// expand + update (one-level backup)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_winner(const int8_t *board, int N, int lane) {
    // search_winner/mod.rs:201-246 on a full board: first cell (row-major) ending a run >= 6
    const int cells = N * N;
    int first = 1 << 20;
    for (int c = lane; c < cells; c += 64) {
        const int v = board[c];
        if (v == 0) continue;
        const int r = c / N, q = c % N;
        const int dr[4] = {0, 1, 1, 1}, dc[4] = {1, 0, 1, -1};
        bool six = false;
#pragma unroll
        for (int dir = 0; dir < 4; dir++) {
            bool ok = true;
#pragma unroll
            for (int k = 1; k < 6; k++) {
                const int rr = r - k * dr[dir], cc = q - k * dc[dir];
                ok = ok && rr >= 0 && cc >= 0 && cc < N && board[rr * N + cc] == v;
            }
            six = six || ok;
        }
        if (six && c < first) first = c;
    }
    first = c6_wave_min_i32(first);
    return first < cells ? (int)board[first] : 0;
}

// expand + update for one game.  A chain of memory round trips, so its shape is dictated by how many of them are serial: everything
// that depends only on g is loaded up front (one trip), the new record / its table entry (slot found in the LDS tags) and the
// last-writer candidates go out behind it, and the backup updates the path's records lane-parallel (they are distinct: levels
// strictly increase along a path).  `prob` = the evaluated prior of the leaf, cells floats (global eval_prior, or the wave's LDS
// scratch inside the fused search kernel); `board` = the leaf position as bytes (terminal leaves only); L.st = the same in registers.
// Returns true when the game's simulation budget for this move is spent (the caller then runs policy()).
// what expand_wave changed of the game's registers (C6GameRegs), for a caller that plays the next round of the same game without reading the
// header back (c6_search_kernel): ok = the round took the ordinary path (nothing but the counters, the allocation mask and possibly the root's record
// changed); took = records allocated (0 for a terminal leaf); left = the allocation page's free mask afterwards; new_root = the root's record if
// this round created it, else -1
struct C6Carry {
    bool ok;
    int took, new_root;
    unsigned long long left;
};
template <int J>
__device__ __forceinline__ bool expand_wave(const C6Dev &E, int g, int lane, const C6Tags &T, const C6GameRegs<J> &R, const C6Leaf<J> &L, float evalue,
                                            const float *prob, const int8_t *board, C6Carry &K) {
    const int cells = E.cells;
    K.ok = true; K.took = 0; K.new_root = -1; K.left = R.alloc_mask;
    const int kind = L.kind, len = L.len, level = L.level, turn = L.turn;
    const unsigned long long key = L.key;
    const int top = R.top, sim_in_move = R.sim_in_move;
    const uint32_t nsearch = R.nsearch;
    float value;
    // these leave together: the rest of the game's words, and the backup's loads (lane i: the path's i-th record)
    const unsigned long long gsims = E.hdr[g].g_sims;
    const uint32_t maxrec = E.hdr[g].g_maxrec;
    const int npages = (int)E.hdr[g].npages, held = E.hdr[g].live_pages;
    const int my_rec = L.my_path & 0xffff, my_cell = (int)((uint32_t)L.my_path >> 16);
    uint32_t bk_slot = 0;
    int bk_visit = 0;
    float bk_qsum = 0.f;
    size_t my_at = 0;
    if (lane < len) {
        my_at = T.rec(my_rec);
        bk_slot = E.rec_stat[my_at].slot;
        bk_visit = E.rec_stat[my_at].visit;
        bk_qsum = E.rec_stat[my_at].qsum;
    }
    const int cell0 = __builtin_amdgcn_readfirstlane(my_cell);      // the root's move of this path (len >= 1): its rootc entry follows the child
    C6RootChild *rootc = E.rootc + (size_t)g * E.cellsP;

    if (kind == C6_LEAF_TERMINAL) {   // mod.rs:338-348: value by winner, no visit increment, no children
        const int w = wave_winner(board, E.N, lane);
        const float cost = (float)turn;
        value = w == turn ? cost : (w != 0 ? -cost : 0.f);
    } else {
        // the record of the newly expanded node: the lowest free one of the allocation page.  A move starts with >= S free records
        // (reprovision_wave), so this cannot run dry; if it ever did, the game stops with an error instead of writing anywhere.
        if (top <= 0 || R.alloc_mask == 0ull) {
            if (lane == 0) E.hdr[g].status = C6_ST_ERROR;
            K.ok = false;
            return false;
        }
        const int bit = __ffsll((long long)R.alloc_mask) - 1;
        const int idx = (R.alloc_page << C6_PAGE_SHIFT) | bit;
        const unsigned long long left = R.alloc_mask & (R.alloc_mask - 1ull);
        const size_t at = T.rec(idx);
        value = evalue;
        float *prior = E.rec_prior + at * (size_t)cells;
        for (int c = lane; c < cells; c += 64) prior[c] = prob[c];      // child.n_prob = prob[r][c], mod.rs:370
        int inserted = 1;
        if (lane == 0) {
            E.hdr[g].free_top = top - 1;
            E.page_free[T.pt[R.alloc_page]] = left;
            const int slot = ht_insert(T, key, idx, 1, 0.f);
            if (slot < 0) { E.hdr[g].status = C6_ST_ERROR; inserted = 0; }
            C6RecStat rs;
            rs.visit = 1;             // parent_node.visit += 1, mod.rs:356 (was 0: never expanded)
            rs.value = value; rs.qsum = 0.f; rs.q = 0.f;
            rs.slot = (uint32_t)(slot < 0 ? 0 : slot);
            rs.level = (int16_t)level; rs.pad = 0;
            rs.key = key;
            uint4 *dst = reinterpret_cast<uint4 *>(E.rec_stat + at);
            const uint4 *src = reinterpret_cast<const uint4 *>(&rs);
            dst[0] = src[0]; dst[1] = src[1];
            if (len == 0) E.hdr[g].root_rec = idx;
            if (len == 1) *reinterpret_cast<uint2 *>(rootc + cell0) = uint2{(uint32_t)(idx + 1) | (1u << 16), 0u};
            const uint32_t used = (uint32_t)(held * C6_PAGE_RECS - (top - 1));
            if (used > maxrec) E.hdr[g].g_maxrec = used;
        }
        K.took = 1;
        K.left = left;
        K.new_root = len == 0 ? idx : -1;
        K.ok = left != 0ull && __builtin_amdgcn_readfirstlane(inserted) != 0;      // (a used-up allocation page: the header names the next one)
        if (left == 0ull && top - 1 > 0) {
            // the allocation page is used up: the next page (in logical order, wrapping) that has a free record.  Once in 64 expansions.
            int next = -1;
            for (int base = 0; base < npages && next < 0; base += 64) {
                const int lp = base + lane;
                const uint32_t ph = lp < npages ? T.pt[lp] : C6_NOPAGE;
                const bool has = ph != C6_NOPAGE && lp != R.alloc_page && E.page_free[ph] != 0ull;
                const unsigned long long m = __ballot(has);
                if (m) next = base + __ffsll((long long)m) - 1;
            }
            if (lane == 0) E.hdr[g].alloc_page = (uint16_t)(next < 0 ? 0 : next);
        }
        // last-writer rule for transposed children (entry().or_insert + n_prob overwrite, mod.rs:366-370): every expanded P with
        // P + i == Q + s sees its child i's prior become prob_Q[s].  i: a stone of the mover's colour the path added to the root
        // (header comment); s: an empty cell of Q; P by key.  The tags answer "absent" for nearly all of them.
        const unsigned long long *zk = E.zob + (turn == C6_WHITE ? C6_MAXCELLS : 0);
        if (len == 1) {
            // Q is a child of the root: i can only be the root's move, the P are the root's other expanded children -- its row has them
            if (R.turn == turn) {
                const uint2 *rc = reinterpret_cast<const uint2 *>(rootc);
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c = lane + 64 * j;
                    if (L.st[j] != 0) continue;
                    const uint32_t pi1 = rc[c].x & 0xffffu;
                    if (pi1 != 0u) E.rec_prior[T.rec((int)pi1 - 1) * (size_t)cells + cell0] = prob[c];
                }
            }
        } else
        for (int base = 0; base < len; base += 64) {
            const int m = base + lane;
            int mcell = my_cell;
            if (base > 0 && m < len) mcell = (int)(E.path_rec[(size_t)g * E.maxd + m] >> 16);
            unsigned long long cand = __ballot(m < len && path_colour(R.turn, R.remain, m) == turn);
            while (cand) {
                const int src_lane = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const int i = __builtin_amdgcn_readlane(mcell, src_lane);
                const unsigned long long kq = key ^ zk[i];
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c = lane + 64 * j;
                    if (L.st[j] != 0) continue;
                    C6HtEntry e;
                    if (ht_find(T, kq ^ zk[c], e)) E.rec_prior[T.rec((int)e.idx1 - 1) * (size_t)cells + i] = prob[c];
                }
            }
        }
    }

    // update, mod.rs:386-408 (one-level backup): lane i owns the path's i-th record (paths beyond 64 steps: further passes); the
    // child statistics in the table entry -- and in the root's row for the root's child -- follow the record
    for (int base = 0; base < len; base += 64) {
        const int i = base + lane;
        if (i < len) {
            const int r = base == 0 ? my_rec : (int)(E.path_rec[(size_t)g * E.maxd + i] & 0xffffu);
            const size_t at = base == 0 ? my_at : T.rec(r);
            const uint32_t slot = base == 0 ? bk_slot : E.rec_stat[at].slot;
            const int v = (base == 0 ? bk_visit : E.rec_stat[at].visit) + 1;
            float qs = base == 0 ? bk_qsum : E.rec_stat[at].qsum;
            if (i == len - 1) {      // the leaf's parent receives the value
                qs = qs + value;
                E.rec_stat[at].qsum = qs;
            }
            const float qv = qs / (float)v;   // recalc_q :77-79
            E.rec_stat[at].visit = v;
            E.rec_stat[at].q = qv;
            C6HtEntry *ep = T.entry(slot);
            ep->visit = sat16(v);
            ep->q = qv;
            if (i == 1) {
                rootc[cell0].visit = sat16(v);
                rootc[cell0].q = qv;
            }
        }
    }
    if (lane == 0) {
        E.hdr[g].nsearch = nsearch + 1;
        E.hdr[g].sim_in_move = sim_in_move + 1;
        E.hdr[g].g_sims = gsims + 1ull;
    }
    return sim_in_move + 1 >= E.S;
}

// ------------------------------------------------------------------------------------------
// move: policy() (mod.rs:433-452) for a game whose simulation budget is spent -> move_pos / move_turn for K1
// ------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ int move_select_wave(const C6Dev &E, int g, int lane, const C6Tags &T, bool publish = true) {
    const int cells = E.cells;
    const int p = E.hdr[g].root_rec;
    const int level = E.hdr[g].nstones;
    const int turn = E.turn[g];
    int pos = -1;
    if (p >= 0 && level < cells) {
        const unsigned long long key = E.hdr[g].root_key;
        const unsigned long long *zk = E.zob + (turn == C6_WHITE ? C6_MAXCELLS : 0);
        const int8_t *board = E.root_board + (size_t)g * E.cellsP;
        int cv[J];
        int vs = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j;
            cv[j] = -1;
            if (c < cells && board[c] == 0) {
                C6HtEntry e;
                cv[j] = 0;
                if (ht_find(T, key ^ zk[c], e)) cv[j] = e.visit == 65535 ? E.rec_stat[T.rec((int)e.idx1 - 1)].visit : (int)e.visit;
                vs += cv[j];
            }
        }
        const float visit_sum = (float)c6_wave_sum_i32(vs);
        float best = -INFINITY;
        float sc[J];
#pragma unroll
        for (int j = 0; j < J; j++) {
            sc[j] = -INFINITY;
            if (cv[j] >= 0) {
                const float visit = (float)cv[j];
                sc[j] = visit / (visit_sum - visit + 1.f);
                best = sc[j] > best ? sc[j] : best;
            }
        }
        best = c6_wave_max_f32(best);
        int mine = -1;   // Iterator::max_by returns the LAST maximum
#pragma unroll
        for (int j = 0; j < J; j++) if (cv[j] >= 0 && sc[j] == best) mine = lane + 64 * j;
        pos = c6_wave_max_i32(mine);
        // pi = root child visit counts (build-side extension)
        uint16_t *pi = E.hist_pi + ((size_t)g * cells + level) * cells;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j;
            if (c < cells) pi[c] = (uint16_t)(cv[j] > 0 ? (cv[j] > 65535 ? 65535 : cv[j]) : 0);
        }
    }
    if (publish && lane == 0) {      // for K1 and the next round's head
        E.move_pos[g] = pos;
        E.move_turn[g] = (int8_t)turn;
    }
    return pos;
}

// ---- pages and chunks: the pools' rings ---------------------------------------------------------------------------------
// pop n ids (n <= 64): lane k < n gets id k; false (nothing taken) if fewer than n are available to this launch.  Ids pushed during
// this launch (by waves on other XCDs, possibly still in their caches) are not ours yet: what a launch may take is fixed when it
// starts (c6_pool_snapshot_kernel) as a number of CREDITS, avail = tail - head.  A popper first takes n credits and only with them in
// hand a ticket from `head` (the earlier add-then-undo on head itself could hand one ring slot to two poppers once an undo
// had landed between two failures).  Credits are conserved: a success removes n for good, a failure removes n for the moment and
// returns them, so the credits handed out never exceed the snapshot and head never passes the snapshot's tail; what a failure in
// flight can cause is only another popper's failure -- that game parks and asks again at the next round's head.
__device__ __forceinline__ bool ring_pop(const uint32_t *ring, uint32_t mask, unsigned long long *ctr, int head_i, int avail_i, int n, int lane, uint32_t &id) {
    unsigned long long h = 0ull;
    int ok = 0;
    if (lane == 0) {
        const long long a = (long long)atomicAdd(&ctr[avail_i], 0ull - (unsigned long long)n);
        ok = a >= (long long)n;
        if (!ok) atomicAdd(&ctr[avail_i], (unsigned long long)n);
        else {
            h = atomicAdd(&ctr[head_i], (unsigned long long)n);
            if (head_i == C6_POOL_PAGE_HEAD) atomicMin(&ctr[C6_POOL_PAGE_LOW], (unsigned long long)(a - (long long)n));
        }
    }
    ok = __shfl(ok, 0, 64);
    if (!ok) return false;
    h = (unsigned long long)__shfl((long long)h, 0, 64);
    id = lane < n ? ring[(uint32_t)(h + (unsigned long long)lane) & mask] : C6_NOPAGE;
    return true;
}
// push the ids of the lanes in `m` (a ballot)
__device__ __forceinline__ void ring_push(uint32_t *ring, uint32_t mask, unsigned long long *ctr, int tail_i, unsigned long long m, int lane, uint32_t id) {
    if (m == 0ull) return;
    unsigned long long t = 0ull;
    if (lane == 0) t = atomicAdd(&ctr[tail_i], (unsigned long long)__popcll(m));
    t = (unsigned long long)__shfl((long long)t, 0, 64);
    if ((m >> lane) & 1ull) ring[(uint32_t)(t + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull))) & mask] = id;
}

// Move boundary (after next()'s prune, at a reset, and for a parked game at every round's head): give game g what its next move may
// need -- at least `want_free` (= E.S) free records and a table of >= 2 x (live + S) slots (a power of two of chunks) -- hand back what
// it does not, and rebuild the table from the live records (its slots hold the records' current statistics).  kill_level: records
// whose level is <= kill_level are dropped first (next(), mod.rs:473-484: the old root's level; 32767: all of them; -1: none).
// want_free = 0 (a finished game that is not recycled): everything goes back.  Returns the record whose key is `find` (-1: none).
// If the pool cannot serve the request, or the game is at the per-game caps, the game is PARKED (status C6_ST_OVERFLOW, hdr.park says
// why) with everything it holds intact and its table rebuilt at the old size, and the call is repeated at the head of the following
// rounds.
template <int J>
__device__ __forceinline__ int reprovision_wave(const C6Dev &E, int g, int lane, int kill_level, unsigned long long find, int want_free) {
    uint32_t *ptg = E.page_table + (size_t)g * E.maxpages, *ckg = E.chunk_table + (size_t)g * E.maxchunks;
    int npages = (int)E.hdr[g].npages;
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    // pass 1: drop the records of the pruned levels; per page its free mask; empty pages are candidates for handing back
    int free_ne = 0, nlive = 0, nempty = 0, nheld = 0;      // wave-uniform; free_ne: free records of the non-empty pages
    for (int base = 0; base < npages; base += 64) {
        const int lp = base + lane;
        const uint32_t ph = lp < npages ? ptg[lp] : C6_NOPAGE;
        unsigned long long my_fm = 0ull;
        const int cnt = npages - base < 64 ? npages - base : 64;
        for (int i = 0; i < cnt; i++) {
            const uint32_t phi = (uint32_t)__builtin_amdgcn_readlane((int)ph, i);
            if (phi == C6_NOPAGE) continue;
            C6RecStat *rs = E.rec_stat + (((size_t)phi << C6_PAGE_SHIFT) | (size_t)lane);
            const int lv = rs->level;
            const bool kill = lv >= 0 && lv <= kill_level;
            if (kill) rs->level = -1;
            const unsigned long long lm = __ballot(lv >= 0 && !kill);
            if (lane == i) my_fm = ~lm;
        }
        const bool valid = ph != C6_NOPAGE;
        if (valid) E.page_free[ph] = my_fm;
        const unsigned long long vm = __ballot(valid), em = __ballot(valid && my_fm == ~0ull);
        nheld += __popcll(vm);
        nempty += __popcll(em);
        const int f = c6_wave_sum_i32(valid && my_fm != ~0ull ? __popcll(my_fm) : 0);
        free_ne += f;
        nlive += C6_PAGE_RECS * __popcll(vm & ~em) - f;
    }
    __threadfence_block();
    // what the next move needs
    const int need_pages = free_ne >= want_free ? 0 : (want_free - free_ne + C6_PAGE_RECS - 1) >> C6_PAGE_SHIFT;      // empty pages to keep / acquire
    const int keep_empty = need_pages < nempty ? need_pages : nempty;
    int acquire = need_pages - keep_empty;
    int Hn = 0;
    if (want_free > 0 || nlive > 0) {
        Hn = C6_CHUNK_SLOTS;
        while (Hn < 2 * (nlive + E.S)) Hn <<= 1;
    }
    int park = C6_PARK_NONE;
    if (nheld - (nempty - keep_empty) + acquire > E.pages_cap || Hn > E.hcap) park = C6_PARK_CAPS;
    // pass 2: hand back the empty pages beyond keep_empty (their logical slots become vacant)
    if (nempty > keep_empty) {
        int seen = 0;
        for (int base = 0; base < npages; base += 64) {
            const int lp = base + lane;
            const uint32_t ph = lp < npages ? ptg[lp] : C6_NOPAGE;
            const bool empty = ph != C6_NOPAGE && E.page_free[ph] == ~0ull;
            const unsigned long long em = __ballot(empty);
            const bool give = empty && seen + __popcll(em & ltmask) >= keep_empty;
            const unsigned long long gm = __ballot(give);
            ring_push(E.page_ring, E.page_ring_mask, E.pool_ctr, C6_POOL_PAGE_TAIL, gm, lane, ph);
            if (give) ptg[lp] = C6_NOPAGE;
            seen += __popcll(em);
        }
        nheld -= nempty - keep_empty;
        __threadfence_block();
    }
    int nfree = free_ne + keep_empty * C6_PAGE_RECS;
    // acquire, in batches of <= 64: pop, then place into the vacant logical slots from the bottom (the caps check above guarantees
    // that enough of them lie below pages_cap)
    if (park == C6_PARK_NONE) {
        int lp0 = 0;
        while (acquire > 0) {
            const int n = acquire < 64 ? acquire : 64;
            uint32_t nid = C6_NOPAGE;
            if (!ring_pop(E.page_ring, E.page_ring_mask, E.pool_ctr, C6_POOL_PAGE_HEAD, C6_POOL_PAGE_AVAIL, n, lane, nid)) { park = C6_PARK_POOL; break; }
            if (lane < n) E.page_free[nid] = ~0ull;      // (its records are level -1: a page is handed back only when it is empty)
            int placed = 0;
            while (placed < n && lp0 < E.pages_cap) {
                const int lp = lp0 + lane;
                const bool vac = lp < E.pages_cap && (lp >= npages || ptg[lp] == C6_NOPAGE);
                const unsigned long long vm = __ballot(vac);
                const int rank = placed + __popcll(vm & ltmask);
                const uint32_t src = (uint32_t)__shfl((int)nid, rank < n ? rank : 0, 64);
                const bool take = vac && rank < n;
                if (take) ptg[lp] = src;
                const int last = c6_wave_max_i32(take ? lp : -1);
                if (last + 1 > npages) npages = last + 1;
                const int avail = __popcll(vm);
                if (avail <= n - placed) { placed += avail; lp0 += 64; }      // group used up
                else placed = n;                                                // the next batch starts in this group again
            }
            __threadfence_block();
            nheld += n;
            nfree += n * C6_PAGE_RECS;
            acquire -= n;
        }
    }
    // trailing vacant slots shrink npages
    {
        int hi = 0;
        for (int base = 0; base < npages; base += 64) {
            const int lp = base + lane;
            const bool valid = lp < npages && ptg[lp] != C6_NOPAGE;
            const int top = c6_wave_max_i32(valid ? lp + 1 : 0);
            if (top > hi) hi = top;
        }
        npages = hi;
    }
    // the table: Hn slots if the game is not parked, else the size it has (a parked game does not search)
    const int Hold = (int)E.hdr[g].hslots;
    if (park != C6_PARK_NONE) Hn = Hold;
    if (Hn != Hold) {
        const int nold = Hold >> C6_CHUNK_SHIFT, nnew = Hn >> C6_CHUNK_SHIFT;
        if (nnew < nold) {
            for (int base = nnew; base < nold; base += 64) {
                const int k = base + lane;
                ring_push(E.chunk_ring, E.chunk_ring_mask, E.pool_ctr, C6_POOL_CHUNK_TAIL, __ballot(k < nold), lane, k < nold ? ckg[k] : 0u);
            }
        } else {
            int have = nold;
            while (have < nnew) {
                const int n = nnew - have < 64 ? nnew - have : 64;
                uint32_t nid = 0u;
                if (!ring_pop(E.chunk_ring, E.chunk_ring_mask, E.pool_ctr, C6_POOL_CHUNK_HEAD, C6_POOL_CHUNK_AVAIL, n, lane, nid)) break;
                if (lane < n) ckg[have + lane] = nid;
                have += n;
            }
            if (have < nnew) {      // not enough chunks: what was taken goes back, the table keeps its old size, the game is parked
                for (int base = nold; base < have; base += 64) {
                    const int k = base + lane;
                    ring_push(E.chunk_ring, E.chunk_ring_mask, E.pool_ctr, C6_POOL_CHUNK_TAIL, __ballot(k < have), lane, k < have ? ckg[k] : 0u);
                }
                park = C6_PARK_CHUNKS;
                Hn = Hold;
            }
        }
    }
    __threadfence_block();
    // rebuild: clear the chunks, insert the live records
    int rr = -1;
    if (Hn > 0) {
        const int nck = Hn >> C6_CHUNK_SHIFT;
        for (int c = 0; c < nck; c++) {
            const size_t cb = (size_t)ckg[c] << C6_CHUNK_SHIFT;
            uint4 *h4 = reinterpret_cast<uint4 *>(E.ht + cb), *t4 = reinterpret_cast<uint4 *>(E.ht_tag + cb);
            for (int i = lane; i < C6_CHUNK_SLOTS; i += 64) h4[i] = uint4{0u, 0u, 0u, 0u};
            t4[lane] = uint4{0u, 0u, 0u, 0u};
        }
        __threadfence_block();
        for (int base = 0; base < npages; base += 64) {
            const int lp = base + lane;
            const uint32_t ph = lp < npages ? ptg[lp] : C6_NOPAGE;
            const int cnt = npages - base < 64 ? npages - base : 64;
            for (int i = 0; i < cnt; i++) {
                const uint32_t phi = (uint32_t)__builtin_amdgcn_readlane((int)ph, i);
                if (phi == C6_NOPAGE) continue;
                const size_t at = ((size_t)phi << C6_PAGE_SHIFT) | (size_t)lane;
                const C6RecStat rs = E.rec_stat[at];
                if (rs.level >= 0) {
                    const int r = ((base + i) << C6_PAGE_SHIFT) | lane;
                    const int slot = ht_insert_atomic(E.ht, E.ht_tag, ckg, Hn, rs.key, r, rs.visit, rs.q);
                    if (slot >= 0) E.rec_stat[at].slot = (uint32_t)slot;
                    if (rs.key == find) rr = r;
                }
            }
        }
    }
    rr = c6_wave_max_i32(rr);
    // the allocation page: the first one with a free record
    int ap = -1;
    for (int base = 0; base < npages && ap < 0; base += 64) {
        const int lp = base + lane;
        const uint32_t ph = lp < npages ? ptg[lp] : C6_NOPAGE;
        const unsigned long long m = __ballot(ph != C6_NOPAGE && E.page_free[ph] != 0ull);
        if (m) ap = base + __ffsll((long long)m) - 1;
    }
    if (lane == 0) {
        C6GameHdr &h = E.hdr[g];
        h.npages = (uint16_t)npages;
        h.alloc_page = (uint16_t)(ap < 0 ? 0 : ap);
        h.hslots = (uint32_t)Hn;
        h.free_top = nfree;
        h.live_pages = nheld;
        h.park = (uint32_t)park;
        if (park != C6_PARK_NONE) { if (h.status == C6_ST_PLAYING) h.status = C6_ST_OVERFLOW; }
        else if (h.status == C6_ST_OVERFLOW) h.status = C6_ST_PLAYING;
    }
    return rr;
}

template <int J>
__device__ __forceinline__ void reset_game_wave(const C6Dev &E, int g, int lane, uint32_t uid, int presims) {
    const int cells = E.cells;
    for (int c = lane; c < cells; c += 64) {
        E.board_soa[(size_t)c * E.Gpad + g] = 0;
        E.root_board[(size_t)g * E.cellsP + c] = 0;
    }
    if (lane == 0) {
        E.turn[g] = C6_BLACK;       // Game::new, game_impl/mod.rs:125-131
        E.remain[g] = 1;
        E.winner[g] = 0;
        E.err[g] = 0;
        E.hdr[g].status = C6_ST_PLAYING;
        E.hdr[g].nstones = 0;
        E.hdr[g].game_uid = uid;
        E.hdr[g].nsearch = 0;
        E.hdr[g].sim_in_move = presims;
        E.hdr[g].root_key = C6_EMPTY_BOARD_KEY;
        E.hdr[g].rootc_ok = 0u;
        E.hdr[g].root_rec = -1;
        E.hdr[g].leaf_kind = C6_LEAF_NONE;
        E.move_pos[g] = -2;
    }
    __threadfence_block();
    reprovision_wave<J>(E, g, lane, 32767, 0ull, E.S);      // every record dropped; pages beyond one move's worth go back to the pool
}

// engine creation: the games own nothing yet (npages = 0, no chunks)
template <int J>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void c6_reset_kernel(C6Dev E) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (g >= E.G) return;
    reset_game_wave<J>(E, g, lane, E.uid_base + (uint32_t)g, 0);
}

// after the stone of game g was placed (by K1, or by the search kernel itself): advance the root, prune (next(), mod.rs:473-487),
// finish / recycle the game.  pos = the cell (-1: the policy had no move), mover / win / err = who placed it and what Game::set +
// search_winner said; leaves move_pos[g] = -2.
template <int J>
__device__ __forceinline__ void post_move_wave(const C6Dev &E, int g, int lane, int pos, int mover, int win_in, int err) {
    const int cells = E.cells;
    const int old_level = E.hdr[g].nstones;
    const unsigned long long old_key = E.hdr[g].root_key;
    unsigned long long new_key = old_key;
    bool finished = false;
    int win = 0;
    int nst = old_level;
    if (lane == 0) { E.move_pos[g] = -2; E.last_move[g] = pos; }

    if (pos < 0) {
        finished = true;                       // policy returned None -> draw, agent_impl/mod.rs:135-137
    } else if (err != 0) {
        if (lane == 0) E.hdr[g].status = C6_ST_ERROR;
        return;
    } else {
        nst = old_level + 1;
        win = win_in;
        finished = win != 0;                   // agent_impl/mod.rs:167-171
        new_key = old_key ^ E.zob[(mover == C6_WHITE ? C6_MAXCELLS : 0) + pos];
        if (lane == 0) {
            E.root_board[(size_t)g * E.cellsP + pos] = (int8_t)mover;
            E.hdr[g].root_key = new_key;
            E.hdr[g].rootc_ok = 0u;      // the next select scans the new root's children and writes their row
            E.moves[(size_t)g * cells + old_level] = (uint16_t)pos;
            E.hdr[g].nstones = (int16_t)nst;
            E.hdr[g].sim_in_move = 0;
            E.hdr[g].g_stones += 1ull;
        }
    }
    __builtin_amdgcn_wave_barrier();

    if (!finished) {
        // prune: nodes of the old root's level are dropped (mod.rs:473-484).  The re-inserted old root (:486) and older levels can
        // never be reached again, so they are dropped too.  Then the pages / table for the next move, and the table rebuilt from the
        // surviving records; rr: record of the new root, if that child was expanded (tree reuse)
        const int rr = reprovision_wave<J>(E, g, lane, old_level, new_key, E.S);
        if (lane == 0) E.hdr[g].root_rec = rr;
        return;
    }

    // game over: PlayResult (agent_impl/mod.rs:41-44) + training tuples (state, pi, turn, z)
    const uint32_t uid = E.hdr[g].game_uid;
    int slot = -1;
    unsigned long long tbase = 0;
    if (lane == 0) {
        slot = (int)atomicAdd(&E.counters[C6_CNT_FETCHQ], 1ull);
        atomicAdd(&E.counters[C6_CNT_FINISHED], 1ull);
        if (E.tuple_buf) tbase = (unsigned long long)atomicAdd(E.tuple_count, (uint32_t)nst);
    }
    slot = __shfl(slot, 0, 64);
    tbase = (unsigned long long)__shfl((long long)tbase, 0, 64);
    if (slot < E.finq_cap) {
        if (lane == 0) { E.fin_winner[slot] = win; E.fin_len[slot] = nst; E.fin_uid[slot] = uid; }
        for (int k = lane; k < nst; k += 64)
            E.fin_moves[(size_t)slot * cells + k] = (int16_t)((pos >= 0 && k == old_level) ? pos : E.moves[(size_t)g * cells + k]);
    }
    if (E.tuple_buf) {
        // replay the game: record k holds the board BEFORE stone k (agent_impl/mod.rs:139-143)
        int8_t st[J];
#pragma unroll
        for (int j = 0; j < J; j++) st[j] = 0;
        int t = C6_BLACK, rem = 1;
        for (int k = 0; k < nst; k++) {
            const unsigned long long rec = tbase + (unsigned long long)k;
            const int mv = (pos >= 0 && k == old_level) ? pos : (int)E.moves[(size_t)g * cells + k];
            if (rec < E.tuple_cap) {
                uint8_t *out = E.tuple_buf + rec * (unsigned long long)E.tuple_stride;
                const uint16_t *pi = E.hist_pi + ((size_t)g * cells + k) * cells;
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c = lane + 64 * j;
                    if (c < cells) {
                        out[c] = (uint8_t)st[j];
                        reinterpret_cast<uint16_t *>(out + cells + (cells & 1))[c] = pi[c];
                    }
                }
                if (lane == 0) {
                    *reinterpret_cast<uint16_t *>(out + E.tuple_stride - 4) = (uint16_t)mv;   // the stone played (Path.pos)
                    out[E.tuple_stride - 2] = (uint8_t)(int8_t)t;
                    out[E.tuple_stride - 1] = (uint8_t)(int8_t)win;
                }
            }
#pragma unroll
            for (int j = 0; j < J; j++) if ((mv >> 6) == j && (mv & 63) == lane) st[j] = (int8_t)t;
            next_turn(t, rem);
        }
        if (lane == 0) atomicAdd(&E.counters[C6_CNT_TUPLES], (unsigned long long)nst);
    }
    if (E.recycle) {
        reset_game_wave<J>(E, g, lane, uid + E.uid_stride, 0);
    } else {
        if (lane == 0) {
            E.hdr[g].status = C6_ST_FINISHED;
            E.hdr[g].root_rec = -1;
            E.winner[g] = (int8_t)win;
        }
        __threadfence_block();
        reprovision_wave<J>(E, g, lane, 32767, 0ull, 0);      // a finished game holds nothing
    }
}

// head of every round for game g: the move K1 applied last round, if any; a parked game asks the pools again (pages come back from
// other games all the time, and the host may have mapped more memory or raised the caps).  Steady-state mode also abandons -- counted,
// never silently -- a slot whose game stopped with NaN scores, and a parked one while the host says that its request cannot be met
// (E.abandon_parked: a mask of C6_PARK_* reasons -- caps at their maximum, a pool that could not grow).
template <int J>
__device__ __forceinline__ void round_head_wave(const C6Dev &E, int g, int lane) {
    const int pos = E.move_pos[g];
    if (pos != -2) {
        int mover = 0, win = 0, err = 0;
        if (pos >= 0) { mover = E.move_turn[g]; win = E.winner[g]; err = E.err[g]; }
        post_move_wave<J>(E, g, lane, pos, mover, win, err);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the root this wave just wrote is read back below
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return;
    }
    int stt = E.hdr[g].status;
    if (stt == C6_ST_OVERFLOW) {
        const int rr = reprovision_wave<J>(E, g, lane, -1, E.hdr[g].root_key, E.S);
        if (lane == 0 && rr >= 0) E.hdr[g].root_rec = rr;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        stt = E.hdr[g].status;
        if (!(stt == C6_ST_OVERFLOW && E.recycle && (E.abandon_parked & (int)E.hdr[g].park))) return;
    }
    if (E.recycle && (stt == C6_ST_ERROR || stt == C6_ST_OVERFLOW)) {
        const uint32_t uid_ = E.hdr[g].game_uid;
        if (lane == 0) atomicAdd(&E.counters[C6_CNT_OVERFLOW], 1ull);
        reset_game_wave<J>(E, g, lane, uid_ + E.uid_stride, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// The rare paths of the search kernel (a move: one round in S; a stone from outside) as real functions that read the engine
// descriptor from a copy in LDS: inlined, their ~20 extra pointers stayed live in scalar registers across the whole kernel, which
// then spilled 113-192 SGPRs and up to 90 VGPRs in its hot loop.
template <int J>
__device__ __noinline__ void round_head_call(const C6Dev *Es, int g, int lane) { round_head_wave<J>(*Es, g, lane); }
// policy() (mod.rs:433-452), then Game::set + search_winner (game_impl/mod.rs:148-169) on the root position, then next()
template <int J>
__device__ __noinline__ void move_call(const C6Dev *Es, int g, int lane, uint8_t *tag_row, int8_t *lboard) {
    const C6Dev &E = *Es;
    C6Tags T;      // the wave's LDS copies (tags_load) are still current: nothing of the game's storage has moved since
    tags_bind(E, tag_row, T);
    T.mask = E.hdr[g].hslots - 1u;
    T.lds = E.hdr[g].hslots <= (uint32_t)C6_TAG_LDS_MAX;
    const int pos = move_select_wave<J>(E, g, lane, T, false);
    const int mover = E.turn[g];
    int win = 0;
    if (pos >= 0) {
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j;
            lboard[c] = c == pos ? (int8_t)mover : (c < E.cells ? E.root_board[(size_t)g * E.cellsP + c] : (int8_t)0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        win = wave_winner(lboard, E.N, lane);
        if (lane == 0) {      // K1's operands follow (canonical state: c6_engine_boards, apply_moves, the other kernels)
            E.board_soa[(size_t)pos * E.Gpad + g] = (int8_t)mover;
            int nt = mover, nr = E.remain[g];
            next_turn(nt, nr);
            E.turn[g] = (int8_t)nt;
            E.remain[g] = (int8_t)nr;
            E.winner[g] = (int8_t)win;
        }
    }
    post_move_wave<J>(E, g, lane, pos, mover, win, 0);
}

// ---- the kernels of a round -----------------------------------------------------------------------------------------------
// Built-in evaluators (RandomEvaluator / uniform: BASELINE configs 1 and 5): `nsims` whole rounds of a game in ONE launch.  A round =
// search() -- select -> evaluate -> expand / backup in registers, the leaf never travels through memory -- and, when the move's
// simulation budget is spent, policy() + Game::set + search_winner + next() (prune / finish / recycle) by the same wave: no K1
// launch, no hand-over through move_pos.  Games do not interact, so a wave runs its game's rounds back to back: from the second
// one on, the game's header, tag row, root row and root priors come out of the L2 instead of HBM.
template <int J, int KIND>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, J <= 2 ? C6_OCC2 : (J <= 4 ? C6_OCC4 : 2)) void c6_search_kernel(C6Dev E, int nsims) {
    __shared__ double gscratch[WAVES_PER_BLOCK][C6_GAM_SCRATCH_DOUBLES(J)];    // c6_wave_gamma hand-back + list, then the leaf's prior / board: wave-private
    __shared__ C6Dev Es;                                    // the descriptor for the rare paths (round_head_call / move_call)
    extern __shared__ __align__(16) uint8_t c6_tag_lds[];   // [WAVES_PER_BLOCK][tag_row_bytes(H)] the games' tag rows
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = blockIdx.x * WAVES_PER_BLOCK + wv;
    {   // E is the kernel's first argument: copied word by word from the kernarg segment (through registers it would cost ~100 SGPRs)
        static_assert(sizeof(C6Dev) % 4 == 0, "copied as dwords");
        typedef __attribute__((address_space(4))) const uint32_t c6_const_u32;
        c6_const_u32 *ka = (c6_const_u32 *)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t *dst = reinterpret_cast<uint32_t *>(&Es);
        for (int i = threadIdx.x; i < (int)(sizeof(C6Dev) / 4); i += 64 * WAVES_PER_BLOCK) dst[i] = ka[i];
    }
    __syncthreads();
    if (g >= E.G) return;
    // one batch of loads: the two words that decide whether there is a move to finish first (a stone from outside,
    // c6_engine_apply_moves; a game abandoned in steady-state mode), and the game's registers
    const int pos0 = E.move_pos[g];
    int stt = E.hdr[g].status;
    C6GameRegs<J> R;
    load_game_regs<J>(E, g, lane, R);
    if (pos0 != -2 || stt == C6_ST_OVERFLOW || (E.recycle && stt == C6_ST_ERROR)) {
        round_head_call<J>(&Es, g, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        stt = E.hdr[g].status;
        load_game_regs<J>(E, g, lane, R);
    }
    if (stt != C6_ST_PLAYING) return;
    C6Tags T;
    uint8_t *tag_row = c6_tag_lds + (size_t)wv * c6_wave_lds(E.hcap, E.pages_cap);
    tags_load(E, g, lane, tag_row, T);
    free_record_prefetch<J>(E, T, R);
    float *probs = reinterpret_cast<float *>(gscratch[wv]);
    int8_t *lboard = reinterpret_cast<int8_t *>(probs + 64 * J);
    C6_STAMP_INIT();
    C6_STAMP(C6_PH_OUT, C6_PH_HEAD);
    for (int it = 0;;) {
        C6Leaf<J> L;
        C6_STAMP(C6_PH_HEAD, C6_PH_STEP);
        select_wave<J>(E, g, lane, gscratch[wv], T, R, L);
        C6_STAMP(C6_PH_STEP, C6_PH_EVAL);
        if (L.kind < 0) { select_failed(E, g, lane, L.kind); return; }
        if (lane == 0) {
            E.hdr[g].g_select += (unsigned long long)L.len;      // per-game statistics: no shared-address atomics
            if ((uint32_t)L.len > E.hdr[g].g_maxdepth) E.hdr[g].g_maxdepth = (uint32_t)L.len;
        }
        float pr[J], value = 0.f;
        if (L.kind == C6_LEAF_EVAL) {
            eval_builtin_regs<J>(E, g, lane, KIND, R.nsearch, L.turn, L.st, pr, value);
#pragma unroll
            for (int j = 0; j < J; j++) probs[lane + 64 * j] = pr[j];
        } else {
#pragma unroll
            for (int j = 0; j < J; j++) lboard[lane + 64 * j] = L.st[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (L.len > 64) __threadfence();      // path steps beyond 64 went through memory
        C6_STAMP(C6_PH_EVAL, C6_PH_EXPAND);
        const bool had_root = R.root_rec >= 0;
        C6Carry K;
        const bool spent = expand_wave<J>(E, g, lane, T, R, L, value, probs, lboard, K);
        C6_STAMP(C6_PH_EXPAND, C6_PH_MOVE);
        if (K.ok && !spent) {
            // The ordinary round (r4): the game belongs to this wave alone, so what the next round needs of the header is known without reading it
            // back -- two dependent memory round trips (the header line, then the allocation page's free mask) per round.  What this round wrote
            // (records, table entries, the root's row) is read by the same wave: wavefront scope orders it, no wait for the stores to land.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            C6_STAMP(C6_PH_MOVE, C6_PH_HEAD);
            if (++it >= nsims) break;
            R.nsearch += 1u;
            R.sim_in_move += 1;
            R.top -= K.took;
            R.alloc_mask = K.left;
            if (K.new_root >= 0) R.root_rec = K.new_root;
            if (had_root) R.rootc_ok = 1;      // select's first scan of a root writes its row (select_wave, d == 0)
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // what this round wrote is read back below / by the next round
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (spent) {
            move_call<J>(&Es, g, lane, tag_row, lboard);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        C6_STAMP(C6_PH_MOVE, C6_PH_HEAD);
        if (++it >= nsims) break;
        // the next round of the same game: its state as this wave left it
        if (E.hdr[g].status != C6_ST_PLAYING) break;
        load_game_regs<J>(E, g, lane, R);
        if (spent) tags_load(E, g, lane, tag_row, T);      // the table was rebuilt, pages may have changed hands
        free_record_prefetch<J>(E, T, R);
    }
    C6_STAMP(C6_PH_HEAD, C6_PH_OUT);
    C6_STAMP_FLUSH();
}

// Net / callback evaluators: the evaluator sits between select and expand, so the leaf travels through the leaf_* arrays.
template <int J>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, J <= 2 ? 4 : (J <= 4 ? 3 : 2)) void c6_select_kernel(C6Dev E) {
    __shared__ double gscratch[WAVES_PER_BLOCK][C6_GAM_SCRATCH_DOUBLES(J)];
    extern __shared__ __align__(16) uint8_t c6_tag_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (g >= E.G) return;
    round_head_wave<J>(E, g, lane);
    if (E.hdr[g].status != C6_ST_PLAYING) {
        if (lane == 0) E.hdr[g].leaf_kind = C6_LEAF_NONE;
        return;
    }
    C6GameRegs<J> R;
    load_game_regs<J>(E, g, lane, R);
    C6Tags T;
    tags_load(E, g, lane, c6_tag_lds + (size_t)wv * c6_wave_lds(E.hcap, E.pages_cap), T);
    C6Leaf<J> L;
    select_wave<J>(E, g, lane, gscratch[wv], T, R, L);
    if (L.kind < 0) { select_failed(E, g, lane, L.kind); return; }
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c = lane + 64 * j;
        if (c < E.cells) E.leaf_board[(size_t)g * E.cellsP + c] = L.st[j];
    }
    if (lane < 64 && lane < L.len) E.path_rec[(size_t)g * E.maxd + lane] = (uint32_t)L.my_path;
    if (lane == 0) {
        E.hdr[g].leaf_kind = (int8_t)L.kind;
        E.hdr[g].leaf_turn = (int8_t)L.turn;
        E.hdr[g].leaf_key = L.key;
        E.hdr[g].leaf_level = (int16_t)L.level;
        E.hdr[g].path_len = L.len;
        E.hdr[g].g_select += (unsigned long long)L.len;      // per-game statistics: no shared-address atomics
        if ((uint32_t)L.len > E.hdr[g].g_maxdepth) E.hdr[g].g_maxdepth = (uint32_t)L.len;
    }
}

template <int J>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void c6_expand_kernel(C6Dev E) {
    extern __shared__ __align__(16) uint8_t c6_tag_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (g >= E.G) return;
    C6Leaf<J> L;
    L.kind = E.hdr[g].leaf_kind;
    L.len = E.hdr[g].path_len;
    L.level = E.hdr[g].leaf_level;
    L.turn = E.hdr[g].leaf_turn;
    L.key = E.hdr[g].leaf_key;
    L.my_path = (int)E.path_rec[(size_t)g * E.maxd + (lane < E.maxd ? lane : 0)];
    const float evalue = E.hdr[g].eval_value;
    C6GameRegs<J> R;
    load_game_regs<J>(E, g, lane, R);      // (some root fields go unused here; one loader keeps both paths on the same code)
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c = lane + 64 * j;
        L.st[j] = c < E.cells ? E.leaf_board[(size_t)g * E.cellsP + c] : (int8_t)2;
    }
    if (L.kind == C6_LEAF_NONE) return;
    C6Tags T;
    tags_load(E, g, lane, c6_tag_lds + (size_t)wv * c6_wave_lds(E.hcap, E.pages_cap), T);
    free_record_prefetch<J>(E, T, R);
    C6Carry K;
    if (expand_wave<J>(E, g, lane, T, R, L, evalue, E.eval_prior + (size_t)g * E.cells, E.leaf_board + (size_t)g * E.cellsP, K)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        move_select_wave<J>(E, g, lane, T);
    }
}

// the stand-alone forms: flush at the end of c6_engine_step (so that callers see every applied move finished) and
// c6_engine_apply_moves (stones from outside)
template <int J>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void c6_post_move_kernel(C6Dev E) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (g >= E.G) return;
    round_head_wave<J>(E, g, lane);
}

// stagger: pretend game g already spent (g mod S) simulations of its first move
__global__ void c6_stagger_kernel(C6Dev E) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < E.G) E.hdr[g].sim_in_move = g % E.S;
}

// limit = tail of both rings: what earlier launches handed back becomes available to this one (c6_engine.h)
__global__ void c6_pool_snapshot_kernel(C6Dev E) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        E.pool_ctr[C6_POOL_PAGE_LIMIT] = E.pool_ctr[C6_POOL_PAGE_TAIL];
        E.pool_ctr[C6_POOL_CHUNK_LIMIT] = E.pool_ctr[C6_POOL_CHUNK_TAIL];
        E.pool_ctr[C6_POOL_PAGE_AVAIL] = E.pool_ctr[C6_POOL_PAGE_TAIL] - E.pool_ctr[C6_POOL_PAGE_HEAD];
        E.pool_ctr[C6_POOL_CHUNK_AVAIL] = E.pool_ctr[C6_POOL_CHUNK_TAIL] - E.pool_ctr[C6_POOL_CHUNK_HEAD];
    }
}
void c6_launch_pool_snapshot(const C6Dev &d, hipStream_t s) { hipLaunchKernelGGL(c6_pool_snapshot_kernel, dim3(1), dim3(64), 0, s, d); }

// Self-test of the rings' pop protocol (c6_pool_ring_selftest, connect6_amd.h): every wave tries `tries` pops of 1-8 ids from a ring that
// holds far fewer than all of them ask for, so that most of the launch runs with the ring nearly dry -- small requests succeeding beside
// larger ones failing.  count[id] is how often an id was handed out.
__global__ void c6_ring_selftest_kernel(const uint32_t *ring, uint32_t mask, unsigned long long *ctr, uint32_t *count, uint32_t *popped, int tries, uint32_t seed) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t x = (wave + 1u) * 2654435761u ^ seed;
    for (int t = 0; t < tries; t++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const int n = (int)((0x84321111u >> ((x & 7u) * 4u)) & 15u);      // 1, 1, 1, 1, 2, 3, 4, 8: many single-id requests beside large ones
        uint32_t id = C6_NOPAGE;
        if (ring_pop(ring, mask, ctr, C6_POOL_PAGE_HEAD, C6_POOL_PAGE_AVAIL, n, lane, id)) {
            if (lane < n) atomicAdd(&count[id & mask], 1u);
            if (lane == 0) atomicAdd(popped, (uint32_t)n);
        }
        if (x >> 31) __builtin_amdgcn_s_sleep(2);
    }
}
void c6_launch_ring_selftest(const uint32_t *ring, uint32_t mask, unsigned long long *ctr, uint32_t *count, uint32_t *popped, int waves, int tries, uint32_t seed, hipStream_t s) {
    hipLaunchKernelGGL(c6_ring_selftest_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, ring, mask, ctr, count, popped, tries, seed);
}

// Games that stopped in an error state leave as they stand: PlayResult{winner None, the path so far} -- what a caller of the
// reference would hold had Agent::play returned its Err (agent_impl/mod.rs:163) instead of the bindings' unwrap().
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void c6_flush_errors_kernel(C6Dev E) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (g >= E.G) return;
    const int stt = E.hdr[g].status;
    if (stt != C6_ST_ERROR && stt != C6_ST_OVERFLOW) return;
    const int nst = E.hdr[g].nstones;
    int slot = -1;
    if (lane == 0) {
        slot = (int)atomicAdd(&E.counters[C6_CNT_FETCHQ], 1ull);
        atomicAdd(&E.counters[C6_CNT_OVERFLOW], 1ull);
    }
    slot = __shfl(slot, 0, 64);
    if (slot < E.finq_cap) {
        if (lane == 0) { E.fin_winner[slot] = 0; E.fin_len[slot] = nst; E.fin_uid[slot] = E.hdr[g].game_uid; }
        for (int k = lane; k < nst; k += 64) E.fin_moves[(size_t)slot * E.cells + k] = (int16_t)E.moves[(size_t)g * E.cells + k];
    }
    if (lane == 0) { E.hdr[g].status = C6_ST_FINISHED; E.winner[g] = 0; }
}
void c6_launch_flush_errors(const C6Dev &d, hipStream_t s) {
    hipLaunchKernelGGL(c6_flush_errors_kernel, dim3((d.G + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(64 * WAVES_PER_BLOCK), 0, s, d);
}

static inline int wave_grid(int G) { return (G + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK; }
static inline size_t tag_lds(const C6Dev &d) { return (size_t)WAVES_PER_BLOCK * c6_wave_lds(d.hcap, d.pages_cap); }      // dynamic LDS of the kernels that hold a game's tag row / lists
size_t c6_tree_lds_bytes(const C6Dev &d) { return tag_lds(d); }

// the dynamic-LDS limit of a kernel is raised (once per kernel and device) when the per-game caps have grown the LDS copies past the default
#include <mutex>
#include <set>
#include <utility>
static void tree_smem(const void *fn, size_t dyn) {
    if (dyn <= 32768) return;
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert({fn, dev}).second) c6_note_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) of a tree kernel");
}
// J = cells owned by one lane: 2 (9x9), 4 (15x15), 6 (19x19)
#define C6_LAUNCH_ONE(KERNEL, d, s, ...)                                                                              \
    do {                                                                                                              \
        tree_smem(reinterpret_cast<const void *>(&KERNEL), tag_lds(d));                                               \
        hipLaunchKernelGGL((KERNEL), dim3(wave_grid((d).G)), dim3(64 * WAVES_PER_BLOCK), tag_lds(d), s, __VA_ARGS__);  \
    } while (0)
#define C6_LAUNCH_J(KERNEL, d, s)                                                                                     \
    do {                                                                                                              \
        const int j_ = ((d).cells + 63) / 64;                                                                         \
        if (j_ <= 2) C6_LAUNCH_ONE(KERNEL<2>, d, s, d);                                                               \
        else if (j_ <= 4) C6_LAUNCH_ONE(KERNEL<4>, d, s, d);                                                          \
        else C6_LAUNCH_ONE(KERNEL<6>, d, s, d);                                                                       \
    } while (0)
#define C6_LAUNCH_JK(KERNEL, K, d, n, s)                                                                              \
    do {                                                                                                              \
        const int j_ = ((d).cells + 63) / 64;                                                                         \
        if (j_ <= 2) C6_LAUNCH_ONE((KERNEL<2, K>), d, s, d, n);                                                       \
        else if (j_ <= 4) C6_LAUNCH_ONE((KERNEL<4, K>), d, s, d, n);                                                  \
        else C6_LAUNCH_ONE((KERNEL<6, K>), d, s, d, n);                                                               \
    } while (0)

void c6_launch_search(const C6Dev &d, int kind, int nsims, hipStream_t s) {      // kind 1: RandomEvaluator, 2: uniform, 3: stress (peaked / uniform by uid)
    c6_launch_pool_snapshot(d, s);
    if (kind == 1) C6_LAUNCH_JK(c6_search_kernel, 1, d, nsims, s);
    else if (kind == 2) C6_LAUNCH_JK(c6_search_kernel, 2, d, nsims, s);
    else C6_LAUNCH_JK(c6_search_kernel, 3, d, nsims, s);
}
void c6_launch_select(const C6Dev &d, hipStream_t s) {
    c6_launch_pool_snapshot(d, s);
    C6_LAUNCH_J(c6_select_kernel, d, s);
}
void c6_launch_expand(const C6Dev &d, hipStream_t s) { C6_LAUNCH_J(c6_expand_kernel, d, s); }
void c6_launch_recover(const C6Dev &d, hipStream_t s) {
    hipLaunchKernelGGL(c6_recover_kernel, dim3(wave_grid(d.G)), dim3(64 * WAVES_PER_BLOCK), 0, s, d);
}
void c6_launch_augment(const C6Dev &d, float *img, hipStream_t s) {
    hipLaunchKernelGGL(c6_augment_kernel, dim3(wave_grid(d.G)), dim3(64 * WAVES_PER_BLOCK), 0, s, d, img);
}
void c6_launch_post_move(const C6Dev &d, hipStream_t s) {
    c6_launch_pool_snapshot(d, s);
    C6_LAUNCH_J(c6_post_move_kernel, d, s);
}
void c6_launch_reset_games(const C6Dev &d, int stagger, hipStream_t s) {
    c6_launch_pool_snapshot(d, s);
    C6_LAUNCH_J(c6_reset_kernel, d, s);
    if (stagger) hipLaunchKernelGGL(c6_stagger_kernel, dim3((d.G + 255) / 256), dim3(256), 0, s, d);
}

// ---- stand-alone operators for the parity tests ---------------------------------------------

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void c6_noise_op_kernel(int N, int num, double alpha, double gd, double gc,
                                                                           const uint32_t *k0, const uint32_t *k1,
                                                                           const uint32_t *search, const uint32_t *depth,
                                                                           const int8_t *boards, float *noise) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (b >= num) return;
    const int cells = N * N;
    const int8_t *board = boards + (size_t)b * cells;
    __shared__ double gscratch[WAVES_PER_BLOCK][C6_GAM_SCRATCH_DOUBLES(C6_MAXJ)];
    c6_gam gam[C6_MAXJ];
    float nz[C6_MAXJ];
    bool need[C6_MAXJ];
#pragma unroll
    for (int j = 0; j < C6_MAXJ; j++) {
        const int c = lane + 64 * j;
        need[j] = c < cells && board[c] == 0;
    }
    // the same wave-cooperative sampler as c6_select_kernel (the oracle draws cell by cell with its own loop)
    c6_wave_gamma<C6_MAXJ>(need, gam, gscratch[threadIdx.x >> 6], lane, alpha, gd, gc, k0[b], k1[b], search[b], depth[b]);
    c6_wave_dirichlet<C6_MAXJ>(gam, nz);
#pragma unroll
    for (int j = 0; j < C6_MAXJ; j++) {
        const int c = lane + 64 * j;
        if (c < cells) noise[(size_t)b * cells + c] = board[c] == 0 ? nz[j] : 0.f;
    }
}

void c6_launch_noise_op(int N, int num, double alpha, double gd, double gc, const uint32_t *k0, const uint32_t *k1,
                        const uint32_t *search, const uint32_t *depth, const int8_t *boards, float *noise, hipStream_t s) {
    hipLaunchKernelGGL(c6_noise_op_kernel, dim3(wave_grid(num)), dim3(64 * WAVES_PER_BLOCK), 0, s, N, num, alpha, gd, gc, k0, k1,
                       search, depth, boards, noise);
}

__global__ void c6_augment_op_kernel(int cells, int num, const int8_t *boards, const int16_t *aug_map, int8_t *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)num * 8 * cells;
    if (i >= total) return;
    const int c = (int)(i % cells), k = (int)((i / cells) % 8);
    const size_t b = i / ((size_t)8 * cells);
    out[i] = boards[b * cells + aug_map[k * cells + c]];
}

void c6_launch_augment_op(int N, int num, const int8_t *boards, const int16_t *aug_map, int8_t *out, hipStream_t s) {
    const size_t total = (size_t)num * 8 * N * N;
    hipLaunchKernelGGL(c6_augment_op_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, N * N, num, boards, aug_map, out);
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void c6_recover_op_kernel(int N, int num, const int8_t *boards, const float *values8,
                                                                             const float *policies8, const int16_t *rec_map,
                                                                             float *value, float *prior) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (b >= num) return;
    const int cells = N * N;
    const float *pol = policies8 + (size_t)b * 8 * cells;
    const float *val = values8 + (size_t)b * 8;
    auto pf = [&](int k, int cell) -> float { return pol[k * cells + cell]; };
    auto vf = [&](int k) -> float { return val[k]; };
    recover_wave(cells, lane, rec_map, boards + (size_t)b * cells, pf, vf, &value[b], prior + (size_t)b * cells);
}

void c6_launch_recover_op(int N, int num, const int8_t *boards, const float *values8, const float *policies8,
                          const int16_t *rec_map, float *value, float *prior, hipStream_t s) {
    hipLaunchKernelGGL(c6_recover_op_kernel, dim3(wave_grid(num)), dim3(64 * WAVES_PER_BLOCK), 0, s, N, num, boards, values8,
                       policies8, rec_map, value, prior);
}

// rollout.hip -- DefaultPolicy, pure MCTS with uniformly random playouts, for thousands of independent
// policies at once.  Restates policy/default_policy/mod.rs:84-275; include/connect6_amd.h part 5 is the contract and
// oracle/c6_default.c the bit-exact CPU restatement the tests compare with.
//
// Mapping.  A search is a chain of dependent steps (select walks the tree the previous search updated, a playout places
// one stone after another), so the parallel axis is the GAME: one thread per policy, 64 policies per wave, no cross-lane
// traffic.  Per thread:
//   - the position being played is two bitboards (Black, White; bit = row * N + col) in LDS, [word][lane] so that the 64
//     lanes of a wave hit 64 different banks; the root position stays in registers and is copied in per search;
//   - a playout step = one Philox word -> k-th empty cell by popcount + c6_select_bit -> set bit -> count the run through
//     the new stone in 4 directions (equal to the reference's full-board search_winner because the position had no
//     winner before the stone);
//   - the tree lives in HBM, private to the slot: open-addressed table  Zobrist key -> node  (entries stamped with the
//     slot's generation, so a reset costs nothing), nodes {visit, black_win, head} and the pushes of next_node as a linked
//     list {cell, child, next} walked newest-first, which turns max_by's "last of equal maxima" into "first strictly
//     greater".  Nothing is ever freed (the reference never prunes this map either).
// Every loop has a bound every lane reaches (edge chains strictly decrease, probes <= table size, playouts <= empty cells).
#include "../../include/connect6_amd.h"
#include "c6_device.h"

#include <algorithm>
#include <string>
#include <vector>

#define C6_TAG_ROLL 6u
#define C6_TAG_ROLL_POLICY 7u

int c6_set_error(int code, const char *msg);   // engine.cpp
#define RCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return c6_set_error(C6_ERR_HIP, (std::string(#x) + ": " + hipGetErrorString(e_)).c_str()); } while (0)

enum { ST_NODES = 0, ST_EDGES = 1, ST_SEARCH = 2, ST_GEN = 3, ST_ERR = 4, ST_UID = 5, ST_WORDS = 8 };

struct RollParams {
    int n, cells, games, num_iter;
    uint32_t k0, k1;
    uint32_t cap;       // nodes and edges per slot
    uint32_t ht_size;   // power of two >= 2 * cap
    const unsigned long long *zob;   // [2][C6_MAXCELLS]
    uint4 *ht;          // [games][ht_size]  {key lo, key hi, node, generation}
    int4 *nodes;        // [games][cap]      {visit, black_win, newest edge + 1, 0}
    int4 *edges;        // [games][cap]      {cell, child node, older edge + 1, 0}
    uint32_t *path;     // [games][cells + 1]
    uint32_t *state;    // [games][ST_WORDS]
    const int8_t *boards;
    const uint8_t *active;
    int32_t *pos;
};

// ---- one thread's position ------------------------------------------------------------------------------------------

template <int NW> struct Position {
    unsigned long long (*wb)[64];   // LDS [2 * NW][64]: rows 0..NW-1 Black, NW..2NW-1 White
    int lane, n, cells;
    int turn, remain, empties;

    __device__ __forceinline__ bool has(int white, int r, int c) const {
        const int idx = r * n + c;
        return (wb[white * NW + (idx >> 6)][lane] >> (idx & 63)) & 1ull;
    }
    __device__ __forceinline__ void place(int cell) {   // Simulate::simulate_in, simulate/mod.rs:262-274 (turn not advanced yet)
        wb[(turn == C6_WHITE ? NW : 0) + (cell >> 6)][lane] |= 1ull << (cell & 63);
        empties--;
    }
    __device__ __forceinline__ void advance() {
        if (--remain <= 0) { remain = 2; turn = -turn; }
    }
    // the stone just placed at `cell` by `turn` closes a run of >= 6 (search_winner/mod.rs:201-246 restricted to that stone)
    __device__ bool wins(int cell) const {
        const int white = turn == C6_WHITE, r0 = cell / n, c0 = cell - r0 * n;
        const int dr[4] = {0, 1, 1, 1}, dc[4] = {1, 0, 1, -1};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int cnt = 1;
            for (int r = r0 + dr[d], c = c0 + dc[d]; cnt < 6 && r < n && c >= 0 && c < n && has(white, r, c); r += dr[d], c += dc[d]) cnt++;
            for (int r = r0 - dr[d], c = c0 - dc[d]; cnt < 6 && r >= 0 && c >= 0 && c < n && has(white, r, c); r -= dr[d], c -= dc[d]) cnt++;
            if (cnt >= 6) return true;
        }
        return false;
    }
    // possible[k]: the k-th empty cell in row-major order (k < empties)
    __device__ __forceinline__ int kth_empty(int k) const {
        int cell = -1;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            unsigned long long e = ~(wb[w][lane] | wb[NW + w][lane]);
            if (w == NW - 1 && (cells & 63)) e &= (1ull << (cells & 63)) - 1ull;
            const int pc = __popcll(e);
            if (cell < 0) {
                if (k < pc) cell = w * 64 + c6_select_bit(e, k);
                else k -= pc;
            }
        }
        return cell;
    }
    // any run of >= 6 on the board (a position handed in from outside); 0 or the colour
    __device__ int any_winner() const {
        for (int cell = 0; cell < cells; cell++) {
            const int r0 = cell / n, c0 = cell - r0 * n;
            int white;
            if (has(0, r0, c0)) white = 0;
            else if (has(1, r0, c0)) white = 1;
            else continue;
            const int dr[4] = {0, 1, 1, 1}, dc[4] = {1, 0, 1, -1};
            for (int d = 0; d < 4; d++) {
                int cnt = 1;
                for (int r = r0 + dr[d], c = c0 + dc[d]; cnt < 6 && r < n && c >= 0 && c < n && has(white, r, c); r += dr[d], c += dc[d]) cnt++;
                if (cnt >= 6) return white ? C6_WHITE : C6_BLACK;
            }
        }
        return C6_NONE;
    }
};

struct Stream {   // the C6RNG words of one search
    uint32_t k0, k1, search, d;
    c6_u4 blk;
    __device__ __forceinline__ uint32_t word() {
        const uint32_t i = d & 3u;
        if (i == 0) blk = c6_philox(d >> 2, 0u, search, C6_TAG_ROLL, k0, k1);
        d++;
        return i == 0 ? blk.x : (i == 1 ? blk.y : (i == 2 ? blk.z : blk.w));
    }
    __device__ __forceinline__ int pick(int len) { return (int)__umulhi(word(), (uint32_t)len); }   // (w * len) >> 32
};

// ---- one slot's map ---------------------------------------------------------------------------------------------------

struct Tree {
    uint4 *ht;
    int4 *nodes, *edges;
    uint32_t mask, gen, cap, n_nodes, n_edges;

    __device__ __forceinline__ static uint32_t slot_of(unsigned long long key) { return (uint32_t)(key ^ (key >> 29)) * 0x9E3779B1u; }
    __device__ int find(unsigned long long key) const {
        uint32_t s = slot_of(key) & mask;
        for (uint32_t i = 0; i <= mask; i++, s = (s + 1) & mask) {
            const uint4 e = ht[s];
            if (e.w != gen) return -1;
            if (e.x == (uint32_t)key && e.y == (uint32_t)(key >> 32)) return (int)e.z;
        }
        return -1;
    }
    // HashMap::insert(key, Node::new(board)): a fresh node under the key, in place when the key is there; -1 = pool full
    __device__ int insert_fresh(unsigned long long key) {
        uint32_t s = slot_of(key) & mask;
        for (uint32_t i = 0; i <= mask; i++, s = (s + 1) & mask) {
            const uint4 e = ht[s];
            if (e.w != gen) break;
            if (e.x == (uint32_t)key && e.y == (uint32_t)(key >> 32)) {
                nodes[e.z] = make_int4(0, 0, 0, 0);
                return (int)e.z;
            }
        }
        if (n_nodes >= cap) return -1;
        const uint32_t idx = n_nodes++;
        nodes[idx] = make_int4(0, 0, 0, 0);
        ht[s] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), idx, gen);
        return (int)idx;
    }
    // DefaultPolicy::select at node `cur` for the player `turn` (mod.rs:119-148): the child edge, or false
    __device__ bool select(int cur, int turn, int &cell, int &child) const {
        int e = nodes[cur].z;
        bool have = false;
        float bestp = 0.f;
        for (uint32_t guard = 0; e != 0 && guard < cap; guard++) {
            const int4 ed = edges[e - 1];
            const int4 ch = nodes[ed.y];
            float p = __fdiv_rn((float)ch.y, 1.f + (float)ch.x);
            if (turn == C6_WHITE) p = -p;
            if (!have || p > bestp) { have = true; bestp = p; cell = ed.x; child = ed.y; }   // newest first + strict > = max_by
            e = ed.z;
        }
        return have && bestp != 0.f;
    }
};

__device__ __forceinline__ void derive_turn(int stones, int &turn, int &remain) {   // game/game_impl/mod.rs:125-169
    if (stones == 0) { turn = C6_BLACK; remain = 1; return; }
    const int t = stones - 1;
    turn = ((t >> 1) & 1) ? C6_BLACK : C6_WHITE;
    remain = 2 - (t & 1);
}

// loads slot g's board into registers (rb / rw), returns its Zobrist key and stone count
template <int NW>
__device__ __forceinline__ unsigned long long load_board(const int8_t *bd, int cells, const unsigned long long *zob,
                                                         unsigned long long (&rb)[NW], unsigned long long (&rw)[NW], int &stones) {
    unsigned long long key = C6_EMPTY_BOARD_KEY;
    stones = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        unsigned long long mb = 0, mw = 0;
        for (int b = 0; b < 64; b++) {
            const int c = w * 64 + b;
            if (c >= cells) break;
            const int v = bd[c];
            if (v == 0) continue;
            if (v < 0) mb |= 1ull << b; else mw |= 1ull << b;
            key ^= zob[(v > 0 ? C6_MAXCELLS : 0) + c];
            stones++;
        }
        rb[w] = mb;
        rw[w] = mw;
    }
    return key;
}

template <int NW>
__global__ __launch_bounds__(64) void c6_rollout_next_kernel(RollParams P) {
    __shared__ unsigned long long wb[2 * NW][64];
    const int lane = threadIdx.x;
    const int g = blockIdx.x * 64 + lane;
    if (g >= P.games) return;   // no barrier below: a wave is the whole workgroup and lanes never exchange data
    uint32_t *st = P.state + (size_t)g * ST_WORDS;
    if ((P.active && !P.active[g]) || st[ST_ERR]) { P.pos[g] = -1; return; }

    unsigned long long rb[NW], rw[NW];
    int stones;
    const unsigned long long root_key = load_board<NW>(P.boards + (size_t)g * P.cells, P.cells, P.zob, rb, rw, stones);
    int root_turn, root_remain;
    derive_turn(stones, root_turn, root_remain);

    Position<NW> ps;
    ps.wb = wb; ps.lane = lane; ps.n = P.n; ps.cells = P.cells;
    auto to_root = [&]() {
#pragma unroll
        for (int w = 0; w < NW; w++) { wb[w][lane] = rb[w]; wb[NW + w][lane] = rw[w]; }
        ps.turn = root_turn; ps.remain = root_remain; ps.empties = P.cells - stones;
    };
    to_root();
    const bool root_done = ps.any_winner() != C6_NONE;

    Tree T;
    T.ht = P.ht + (size_t)g * P.ht_size;
    T.nodes = P.nodes + (size_t)g * P.cap;
    T.edges = P.edges + (size_t)g * P.cap;
    T.mask = P.ht_size - 1; T.cap = P.cap; T.gen = st[ST_GEN];
    T.n_nodes = st[ST_NODES]; T.n_edges = st[ST_EDGES];
    uint32_t *path = P.path + (size_t)g * (P.cells + 1);
    const uint32_t uid = st[ST_UID];
    Stream rng;
    rng.k0 = P.k0; rng.k1 = P.k1 + uid;
    uint32_t nsearch = st[ST_SEARCH];
    bool overflow = false;

    int root = T.find(root_key);
    for (int it = 0; it < P.num_iter && !overflow; it++) {
        rng.search = nsearch++; rng.d = 0;
        if (root < 0) {   // init, mod.rs:110-113
            root = T.insert_fresh(root_key);
            if (root < 0) { overflow = true; break; }
        }
        if (root_done) continue;   // select finds no child of a finished position, then mod.rs:224-226 returns
        if (it) to_root();
        int cur = root, np = 0, cell = 0, child = 0;
        unsigned long long key = root_key;
        path[0] = (uint32_t)root;
        bool won = false;
        while (T.select(cur, ps.turn, cell, child)) {   // mod.rs:219-223
            ps.place(cell);
            won = ps.wins(cell);
            key ^= P.zob[(ps.turn == C6_WHITE ? C6_MAXCELLS : 0) + cell];
            cur = child;
            path[++np] = (uint32_t)cur;
            if (won) break;   // a finished node has no children: the next select stops and search returns
            ps.advance();
        }
        if (won || ps.empties == 0) continue;
        // expand, mod.rs:150-170
        cell = ps.kth_empty(rng.pick(ps.empties));
        child = T.insert_fresh(key ^ P.zob[(ps.turn == C6_WHITE ? C6_MAXCELLS : 0) + cell]);
        if (child < 0 || T.n_edges >= T.cap) { overflow = true; break; }
        {
            const uint32_t e = T.n_edges++;
            int4 par = T.nodes[cur];
            T.edges[e] = make_int4(cell, child, par.z, 0);
            par.z = (int)e + 1;
            T.nodes[cur] = par;
        }
        ps.place(cell);
        path[++np] = (uint32_t)child;
        int winner = C6_NONE;
        if (ps.wins(cell)) winner = ps.turn; else ps.advance();
        // update, mod.rs:172-211: random playout, then (1, winner == Black) onto the whole path
        while (winner == C6_NONE && ps.empties > 0) {
            cell = ps.kth_empty(rng.pick(ps.empties));
            ps.place(cell);
            if (ps.wins(cell)) winner = ps.turn; else ps.advance();
        }
        if (winner == C6_NONE) continue;
        const int bw = winner == C6_BLACK;
        for (int i = 0; i <= np; i++) {
            int4 nd = T.nodes[path[i]];
            nd.x += 1; nd.y += bw;
            T.nodes[path[i]] = nd;
        }
    }

    st[ST_NODES] = T.n_nodes; st[ST_EDGES] = T.n_edges; st[ST_SEARCH] = nsearch;
    if (overflow) { st[ST_ERR] = 1; P.pos[g] = -1; return; }
    // policy, mod.rs:238-258
    int out = -1;
    if (root >= 0) {
        to_root();
        int cell = 0, child = 0;
        if (!root_done && T.select(root, ps.turn, cell, child)) out = cell;
        else if (ps.empties > 0) {
            const uint32_t w = c6_philox(0u, 0u, nsearch, C6_TAG_ROLL_POLICY, rng.k0, rng.k1).x;
            out = ps.kth_empty((int)__umulhi(w, (uint32_t)ps.empties));
        }
    }
    P.pos[g] = out;
}

// playouts only (bench): from every board to the end with uniformly random moves
template <int NW>
__global__ __launch_bounds__(64) void c6_rollout_playout_kernel(RollParams P, uint32_t launch, int8_t *winners, unsigned long long *steps) {
    __shared__ unsigned long long wb[2 * NW][64];
    const int lane = threadIdx.x;
    const int g = blockIdx.x * 64 + lane;
    if (g >= P.games) return;
    unsigned long long rb[NW], rw[NW];
    int stones;
    load_board<NW>(P.boards + (size_t)g * P.cells, P.cells, P.zob, rb, rw, stones);
    Position<NW> ps;
    ps.wb = wb; ps.lane = lane; ps.n = P.n; ps.cells = P.cells;
#pragma unroll
    for (int w = 0; w < NW; w++) { wb[w][lane] = rb[w]; wb[NW + w][lane] = rw[w]; }
    derive_turn(stones, ps.turn, ps.remain);
    ps.empties = P.cells - stones;
    Stream rng;
    rng.k0 = P.k0; rng.k1 = P.k1 + (uint32_t)g; rng.search = launch; rng.d = 0;
    int winner = ps.any_winner();
    unsigned long long placed = 0;
    while (winner == C6_NONE && ps.empties > 0) {
        const int cell = ps.kth_empty(rng.pick(ps.empties));
        ps.place(cell);
        placed++;
        if (ps.wins(cell)) winner = ps.turn; else ps.advance();
    }
    if (winners) winners[g] = (int8_t)winner;
    atomicAdd(steps, placed);
}

// test access: the node of `board` in slot g and its children in push order
__global__ void c6_rollout_node_kernel(RollParams P, int g, const int8_t *board, int cap, int32_t *out /* [4 + 3 * cap] */) {
    if (threadIdx.x || blockIdx.x) return;
    unsigned long long key = C6_EMPTY_BOARD_KEY;
    for (int c = 0; c < P.cells; c++)
        if (board[c]) key ^= P.zob[(board[c] > 0 ? C6_MAXCELLS : 0) + c];
    const uint32_t *st = P.state + (size_t)g * ST_WORDS;
    Tree T;
    T.ht = P.ht + (size_t)g * P.ht_size;
    T.nodes = P.nodes + (size_t)g * P.cap;
    T.edges = P.edges + (size_t)g * P.cap;
    T.mask = P.ht_size - 1; T.cap = P.cap; T.gen = st[ST_GEN];
    T.n_nodes = st[ST_NODES]; T.n_edges = st[ST_EDGES];
    const int idx = T.find(key);
    out[0] = idx >= 0;
    if (idx < 0) return;
    const int4 nd = T.nodes[idx];
    out[1] = nd.x; out[2] = nd.y;
    int cnt = 0;
    for (int e = nd.z; e != 0 && cnt < (int)T.cap; e = T.edges[e - 1].z) cnt++;
    out[3] = cnt;
    int i = cnt - 1;
    for (int e = nd.z; e != 0 && i >= 0; e = T.edges[e - 1].z, i--) {
        if (i >= cap) continue;
        const int4 ed = T.edges[e - 1];
        const int4 ch = T.nodes[ed.y];
        out[4 + 3 * i] = ed.x; out[5 + 3 * i] = ch.x; out[6 + 3 * i] = ch.y;
    }
}

__global__ void c6_rollout_reset_kernel(uint32_t *state, int games, const uint8_t *mask, int first) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= games || (mask && !mask[g])) return;
    uint32_t *st = state + (size_t)g * ST_WORDS;
    st[ST_NODES] = 0; st[ST_EDGES] = 0; st[ST_SEARCH] = 0; st[ST_ERR] = 0;
    if (first) { st[ST_GEN] = 1; st[ST_UID] = (uint32_t)g; }
    else { st[ST_GEN] += 1; st[ST_UID] += (uint32_t)games; }
}

// ---- host ---------------------------------------------------------------------------------------------------------------

struct c6_rollout {
    RollParams P{};
    int device = 0, nw = 0;
    hipStream_t stream = nullptr;
    int8_t *boards_dev = nullptr;
    uint8_t *active_dev = nullptr;
    int32_t *pos_dev = nullptr;
    int32_t *query_dev = nullptr;
    int query_cap = 0;
    unsigned long long *steps_dev = nullptr;
    std::vector<void *> owned;
};

static uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

template <typename T> static hipError_t dalloc(c6_rollout *r, T **p, size_t count, bool zero) {
    void *q = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return e;
    if (zero && (e = hipMemsetAsync(q, 0, bytes, r->stream)) != hipSuccess) { hipFree(q); return e; }   // ordered before the kernels of r->stream
    r->owned.push_back(q);
    *p = reinterpret_cast<T *>(q);
    return hipSuccess;
}

extern "C" void c6_rollout_destroy(c6_rollout *r) {
    if (!r) return;
    hipSetDevice(r->device);
    if (r->stream) { hipStreamSynchronize(r->stream); hipStreamDestroy(r->stream); }
    for (void *p : r->owned) hipFree(p);
    delete r;
}

extern "C" int c6_rollout_create(int n, int num_games, int num_iter, uint64_t seed, size_t max_nodes, int device, c6_rollout **out) {
    if (!out || (n != 9 && n != 15 && n != 19) || num_games <= 0 || num_iter < 0)
        return c6_set_error(C6_ERR_BAD_ARG, "c6_rollout_create: board_size must be 9, 15 or 19, num_games > 0, num_iter >= 0");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return c6_set_error(C6_ERR_NO_GPU, "no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= nd) return c6_set_error(C6_ERR_BAD_ARG, "c6_rollout_create: no such device");
    const size_t cells = (size_t)n * n;
    if (max_nodes == 0) max_nodes = (size_t)num_iter * cells + cells + 1;
    if (max_nodes < 2 || max_nodes > (1u << 30)) return c6_set_error(C6_ERR_BAD_ARG, "c6_rollout_create: max_nodes out of range");
    RCHK(hipSetDevice(device));
    c6_rollout *r = new c6_rollout();
    r->device = device;
    r->nw = (int)((cells + 63) / 64);
    RollParams &P = r->P;
    P.n = n; P.cells = (int)cells; P.games = num_games; P.num_iter = num_iter;
    P.k0 = (uint32_t)seed; P.k1 = (uint32_t)(seed >> 32);
    P.cap = (uint32_t)max_nodes;
    uint32_t hs = 16;
    while (hs < 2 * P.cap) hs <<= 1;
    P.ht_size = hs;
    std::vector<unsigned long long> zob(2 * C6_MAXCELLS);
    uint64_t s = 0xD1CE5EEDULL;
    for (auto &v : zob) v = splitmix(s);
    hipError_t e = hipSuccess;
    unsigned long long *zd = nullptr;
    const size_t G = (size_t)num_games;
    if ((e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = dalloc(r, &zd, zob.size(), false)) != hipSuccess ||
        (e = hipMemcpy(zd, zob.data(), zob.size() * sizeof(unsigned long long), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = dalloc(r, &P.ht, G * P.ht_size, true)) != hipSuccess ||       // generation 0 = empty everywhere
        (e = dalloc(r, &P.nodes, G * P.cap, false)) != hipSuccess ||
        (e = dalloc(r, &P.edges, G * P.cap, false)) != hipSuccess ||
        (e = dalloc(r, &P.path, G * (cells + 1), false)) != hipSuccess ||
        (e = dalloc(r, &P.state, G * ST_WORDS, true)) != hipSuccess ||
        (e = dalloc(r, &r->boards_dev, G * cells, true)) != hipSuccess ||
        (e = dalloc(r, &r->active_dev, G, true)) != hipSuccess ||
        (e = dalloc(r, &r->pos_dev, G, true)) != hipSuccess ||
        (e = dalloc(r, &r->steps_dev, 1, true)) != hipSuccess) {
        const std::string msg = std::string("c6_rollout_create: ") + hipGetErrorString(e);
        c6_rollout_destroy(r);
        return c6_set_error(C6_ERR_HIP, msg.c_str());
    }
    P.zob = zd;
    c6_rollout_reset_kernel<<<(num_games + 255) / 256, 256, 0, r->stream>>>(P.state, num_games, nullptr, 1);
    RCHK(hipStreamSynchronize(r->stream));
    *out = r;
    return C6_OK;
}

extern "C" int c6_rollout_reset(c6_rollout *r, const uint8_t *mask) {
    if (!r) return c6_set_error(C6_ERR_BAD_ARG, "c6_rollout_reset: null handle");
    RCHK(hipSetDevice(r->device));
    if (mask) RCHK(hipMemcpyAsync(r->active_dev, mask, (size_t)r->P.games, hipMemcpyHostToDevice, r->stream));
    c6_rollout_reset_kernel<<<(r->P.games + 255) / 256, 256, 0, r->stream>>>(r->P.state, r->P.games, mask ? r->active_dev : nullptr, 0);
    RCHK(hipGetLastError());
    RCHK(hipStreamSynchronize(r->stream));
    return C6_OK;
}

template <typename F> static bool by_words(int nw, F &&f) {
    switch (nw) {
    case 2: f(std::integral_constant<int, 2>{}); return true;   // N = 9
    case 4: f(std::integral_constant<int, 4>{}); return true;   // N = 15
    case 6: f(std::integral_constant<int, 6>{}); return true;   // N = 19
    }
    return false;
}

extern "C" int c6_rollout_next(c6_rollout *r, const int8_t *boards, const uint8_t *active, int32_t *pos, int on_device) {
    if (!r || !boards || !pos) return c6_set_error(C6_ERR_BAD_ARG, "c6_rollout_next: null argument");
    RCHK(hipSetDevice(r->device));
    RollParams P = r->P;
    const size_t G = (size_t)P.games;
    if (on_device) {
        P.boards = boards; P.active = active; P.pos = pos;
    } else {
        RCHK(hipMemcpyAsync(r->boards_dev, boards, G * P.cells, hipMemcpyHostToDevice, r->stream));
        if (active) RCHK(hipMemcpyAsync(r->active_dev, active, G, hipMemcpyHostToDevice, r->stream));
        P.boards = r->boards_dev; P.active = active ? r->active_dev : nullptr; P.pos = r->pos_dev;
    }
    const unsigned blocks = (unsigned)((G + 63) / 64);
    by_words(r->nw, [&](auto nw) { c6_rollout_next_kernel<decltype(nw)::value><<<blocks, 64, 0, r->stream>>>(P); });
    RCHK(hipGetLastError());
    if (!on_device) RCHK(hipMemcpyAsync(pos, r->pos_dev, G * sizeof(int32_t), hipMemcpyDeviceToHost, r->stream));
    RCHK(hipStreamSynchronize(r->stream));
    // a full tree is an error of the call, not a silent -1
    std::vector<uint32_t> st(G * ST_WORDS);
    RCHK(hipMemcpy(st.data(), r->P.state, st.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t g = 0; g < G; g++)
        if (st[g * ST_WORDS + ST_ERR] && (!active || on_device || active[g]))
            return c6_set_error(C6_ERR_OVERFLOW, ("c6_rollout_next: the tree of slot " + std::to_string(g) + " is full (max_nodes = " +
                                                  std::to_string(P.cap) + "); reset the slot or create with a larger max_nodes").c_str());
    return C6_OK;
}

extern "C" int c6_rollout_slot_info(c6_rollout *r, int g, c6_rollout_info *out) {
    if (!r || !out || g < 0 || g >= r->P.games) return c6_set_error(C6_ERR_BAD_ARG, "c6_rollout_slot_info: bad argument");
    RCHK(hipSetDevice(r->device));
    uint32_t st[ST_WORDS];
    RCHK(hipMemcpy(st, r->P.state + (size_t)g * ST_WORDS, sizeof st, hipMemcpyDeviceToHost));
    out->nodes = st[ST_NODES]; out->edges = st[ST_EDGES]; out->searches = st[ST_SEARCH]; out->uid = st[ST_UID];
    return C6_OK;
}

extern "C" int c6_rollout_node(c6_rollout *r, int g, const int8_t *board, int *found, int *visit, int *black_win, int *n_children, int cap,
                               int32_t *child_cell, int32_t *child_visit, int32_t *child_black_win) {
    if (!r || !board || !found || g < 0 || g >= r->P.games || cap < 0) return c6_set_error(C6_ERR_BAD_ARG, "c6_rollout_node: bad argument");
    RCHK(hipSetDevice(r->device));
    if (!r->query_dev || r->query_cap < cap) {
        int32_t *q = nullptr;
        RCHK(dalloc(r, &q, (size_t)4 + 3 * (size_t)cap, true));
        r->query_dev = q; r->query_cap = cap;
    }
    RCHK(hipMemcpyAsync(r->boards_dev, board, (size_t)r->P.cells, hipMemcpyHostToDevice, r->stream));
    c6_rollout_node_kernel<<<1, 64, 0, r->stream>>>(r->P, g, r->boards_dev, cap, r->query_dev);
    RCHK(hipGetLastError());
    std::vector<int32_t> h((size_t)4 + 3 * (size_t)cap);
    RCHK(hipMemcpyAsync(h.data(), r->query_dev, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, r->stream));
    RCHK(hipStreamSynchronize(r->stream));
    *found = h[0];
    if (!h[0]) return C6_OK;
    if (visit) *visit = h[1];
    if (black_win) *black_win = h[2];
    if (n_children) *n_children = h[3];
    for (int i = 0; i < std::min(cap, h[3]); i++) {
        if (child_cell) child_cell[i] = h[4 + 3 * i];
        if (child_visit) child_visit[i] = h[5 + 3 * i];
        if (child_black_win) child_black_win[i] = h[6 + 3 * i];
    }
    return C6_OK;
}

extern "C" int c6_rollout_bench_playouts(c6_rollout *r, const int8_t *boards, int iters, double *avg_ms, double *steps, int8_t *winners) {
    if (!r || !boards || iters <= 0 || !avg_ms) return c6_set_error(C6_ERR_BAD_ARG, "c6_rollout_bench_playouts: bad argument");
    RCHK(hipSetDevice(r->device));
    RollParams P = r->P;
    const size_t G = (size_t)P.games;
    RCHK(hipMemcpyAsync(r->boards_dev, boards, G * P.cells, hipMemcpyHostToDevice, r->stream));
    P.boards = r->boards_dev;
    int8_t *wd = reinterpret_cast<int8_t *>(r->active_dev);
    const unsigned blocks = (unsigned)((G + 63) / 64);
    hipEvent_t e0, e1;
    RCHK(hipEventCreate(&e0));
    RCHK(hipEventCreate(&e1));
    auto launch = [&](uint32_t i) {
        by_words(r->nw, [&](auto nw) { c6_rollout_playout_kernel<decltype(nw)::value><<<blocks, 64, 0, r->stream>>>(P, i, wd, r->steps_dev); });
    };
    launch(0xFFFFFFFFu);   // untimed
    RCHK(hipMemsetAsync(r->steps_dev, 0, sizeof(unsigned long long), r->stream));
    RCHK(hipEventRecord(e0, r->stream));
    for (int i = 0; i < iters; i++) launch((uint32_t)i);
    RCHK(hipEventRecord(e1, r->stream));
    RCHK(hipStreamSynchronize(r->stream));
    RCHK(hipGetLastError());
    float ms = 0.f;
    RCHK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    unsigned long long total = 0;
    RCHK(hipMemcpy(&total, r->steps_dev, sizeof total, hipMemcpyDeviceToHost));
    *avg_ms = (double)ms / iters;
    if (steps) *steps = (double)total / iters;
    if (winners) RCHK(hipMemcpy(winners, wd, G, hipMemcpyDeviceToHost));
    return C6_OK;
}

// c6_engine.h -- HBM layout of the self-play engine and the kernel launchers.
//
// One engine = one GPU = G concurrent games.  Canonical game state is the
// structure-of-arrays board  board_soa[cell][game]  (int8: -1 black, 0 empty, +1 white;
// game/player/mod.rs:8-12) that the board kernel K1 reads with lane == game.  The tree
// kernels run one wave per game and keep, per game:
//   * an AoS copy of the root position (root_board, root_key, root_sig),
//   * "expanded node" records (edge-major: prior[N*N] per record) in PAGES of 64 taken from one pool shared by all games
//     (page table per game; pages are acquired and handed back at move boundaries) and an open-addressed table
//     board key -> record  (64-bit Zobrist) in CHUNKS of 1024 slots from a second pool, which together reproduce the
//     reference's HashMap<u64, Node> (alphazero_policy/mod.rs:186-190): it grows per game, without a bound other than
//     memory, and a game's share is proportional to its tree,
//   * one tag byte per table slot (the wave holds the whole row in LDS: an absent child costs no memory access) and the
//     root's children as one contiguous row (rootc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct C6HtEntry {       // 16 bytes: one load brings everything select needs to know about a child
    unsigned long long key;   // 0 = empty slot
    uint16_t idx1;            // record index + 1
    uint16_t visit;           // the record's visit count, saturating at 65535 (then rec_visit is read)
    float q;                  // the record's q_value
};

// tag of a key in ht_tag (0 = empty slot): the key's top byte -- independent of the home slot, which uses the low bits
#define C6_TAG_LDS_MAX 8192      // tables up to this many slots have their tag row in LDS; larger ones are read in memory
static __host__ __device__ inline uint8_t c6_ht_tag_of(unsigned long long key) { const uint8_t t = (uint8_t)(key >> 56); return t ? t : (uint8_t)1; }

// a child of the root as select reads it: the root is the one node whose children have no other parent on their level (a node's level
// is its stone count), so this copy can be kept exact -- written when the child is expanded and at every backup through it
struct C6RootChild {
    uint16_t idx1;            // record index + 1, 0 = not expanded
    uint16_t visit;           // saturating at 65535
    float q;
};

// ---- paged storage of the trees ---------------------------------------------------------------------------------------------
// Records live in pages of C6_PAGE_RECS: record r of a game is slot r & 63 of pool page page_table[game][r >> 6]; a game's logical
// page slots are stable (a record's index never changes), slots whose page was handed back hold C6_NOPAGE.  The table of a game has
// hslots = C6_CHUNK_SLOTS * 2^k slots: slot s is entry s & 1023 of pool chunk chunk_table[game][s >> 10].  Both pools keep their
// free ids in a ring (tickets head / tail); ids pushed during a kernel become poppable when the NEXT kernel starts (limit = tail is
// taken by c6_pool_snapshot_kernel between launches): a page changes hands across a kernel boundary only, where the caches of all
// XCDs are written back and invalidated -- never between two waves of one launch.
#define C6_PAGE_SHIFT 6
#define C6_PAGE_RECS 64
#define C6_CHUNK_SHIFT 10
#define C6_CHUNK_SLOTS 1024
#define C6_NOPAGE 0xFFFFFFFFu
#define C6_MAX_RECORDS 65000      // per game: record indices travel as 16 bits (idx + 1)
enum { C6_POOL_PAGE_HEAD = 0, C6_POOL_PAGE_TAIL, C6_POOL_PAGE_LIMIT, C6_POOL_CHUNK_HEAD, C6_POOL_CHUNK_TAIL, C6_POOL_CHUNK_LIMIT,
       C6_POOL_PAGE_LOW, C6_POOL_PAGE_AVAIL, C6_POOL_CHUNK_AVAIL, C6_POOL_N };      // PAGE_LOW: fewest free pages seen by a popper (statistics);
                                                                                  // *_AVAIL: the credits of this launch (signed; tree_kernels.hip ring_pop)
enum { C6_PARK_NONE = 0, C6_PARK_POOL = 1, C6_PARK_CAPS = 2, C6_PARK_CHUNKS = 4 };      // bits: C6Dev.abandon_parked is a mask of them

// C6_ST_OVERFLOW: the game is PARKED at a move boundary -- after a move / a reset it could not be given the records (>= S free ones)
// or the table (>= 2 x (live records + S) slots) its next move may need: the pool had no free page (hdr.park = C6_PARK_POOL; other
// games hand pages back all the time, and the host can map more memory) or the game is at the per-game caps (C6_PARK_CAPS; the host
// raises them).  Nothing of a search is ever interrupted: the head of every round retries, and the game plays on as if nothing had been.
// C6_ST_ERROR: NaN scores (the reference panics, mod.rs:287) or an illegal stone from outside (Game::set Err).
enum { C6_ST_PLAYING = 0, C6_ST_FINISHED = 1, C6_ST_ERROR = 2, C6_ST_OVERFLOW = 3 };
enum { C6_LEAF_NONE = 0, C6_LEAF_EVAL = 1, C6_LEAF_TERMINAL = 2 };
enum { C6_CNT_SIMS = 0, C6_CNT_SELECT, C6_CNT_STONES, C6_CNT_FINISHED, C6_CNT_TUPLES, C6_CNT_OVERFLOW, C6_CNT_MAXDEPTH,
       C6_CNT_MAXREC, C6_CNT_LIVE, C6_CNT_FETCHQ, C6_CNT_N };

// Everything a round reads and writes about a game, in ONE 128-byte line: a wave fetches it with a few adjacent loads and lane 0 writes
// it back into the same line (as ~25 separate arrays every word cost a memory transaction of its own: the tree kernels are bound by
// small HBM transactions, profiles/r02_tree_traffic.txt).  K1's operands (board_soa, turn, remain, winner, err, move_pos) stay
// structure-of-arrays over games: K1 reads them with lane == game.
struct C6GameHdr {
    unsigned long long root_key;      //   0  Zobrist key of the root position
    int32_t root_rec;                 //   8  record of the root, -1 = not expanded
    uint32_t game_uid;                //  12  C6RNG key of the game
    uint32_t rootc_ok;                //  16  the rootc row holds the current root's children
    uint16_t npages;                  //  20  logical page slots in use (high-water mark; slots handed back hold C6_NOPAGE)
    uint16_t alloc_page;              //  22  logical page the next record comes from: has a free record whenever free_top > 0
    uint32_t hslots;                  //  24  slots of the game's table: C6_CHUNK_SLOTS * a power of two
    uint32_t park;                    //  28  C6_PARK_*: why the game is in C6_ST_OVERFLOW
    uint32_t nsearch;                 //  32  searches since the game's start
    int32_t sim_in_move;              //  36
    int32_t free_top;                 //  40  free records in the game's pages
    int32_t live_pages;               //  44  pages the game holds (statistics)
    int16_t nstones;                  //  48
    int8_t status;                    //  50  C6_ST_*
    int8_t leaf_kind;                 //  51  C6_LEAF_*
    uint32_t g_maxdepth, g_maxrec;    //  52  statistics of the game slot
    int16_t leaf_level;               //  60
    int8_t leaf_turn, pad0;           //  62
    unsigned long long g_sims, g_select, g_stones;      //  64
    unsigned long long leaf_key;      //  88  leaf exchange (select -> evaluator -> expand)
    uint32_t pad3[4];                 //  96
    int32_t path_len;                 // 112
    float eval_value;                 // 116
    uint32_t pad1[2];                 // 120
};
static_assert(sizeof(C6GameHdr) == 128, "one line per game");

// an expanded node, one 32-byte piece per record (creation and backup touch one sector); its priors are in rec_prior
struct C6RecStat {
    int32_t visit;
    float value, qsum, q;
    uint32_t slot;          // table slot of the record
    int16_t level;          // stones on the board of the node, -1 = free record
    uint16_t pad;
    unsigned long long key; // Zobrist key of the position
};
static_assert(sizeof(C6RecStat) == 32, "one sector per record");

struct C6Dev {
    // geometry / hyper-parameters (policy/alphazero_policy/mod.rs:91-96)
    int N, cells, cellsP, G, Gpad, S, maxd;
    // paged storage: absolute per-game limits (array widths), the caps games may grow to without the host (they size the LDS copies
    // of a game's page table / chunk list / tag row and are raised by c6_engine_maintain), and the pools' sizes
    int maxpages, maxchunks, pages_cap, hcap;
    unsigned int pages_total, chunks_total, page_ring_mask, chunk_ring_mask;
    int stress_every;        // C6_EVAL_STRESS: games with uid % stress_every == 0 get the peaked evaluator
    int abandon_parked;      // steady-state mode: mask of C6_PARK_* -- a game parked for one of these reasons is abandoned (counted) at the next round's
                             // head: the host sets a bit while that request cannot be met (caps at their maximum; a pool that could not grow at the last
                             // c6_engine_maintain) and clears it when it can again
    float eps, cpuct;
    double alpha, gam_d, gam_c;
    uint32_t seed_lo, seed_hi, uid_base, uid_stride;
    int recycle;
    // canonical game state (SoA over games)
    int8_t *board_soa;     // [cells][Gpad]
    int8_t *turn, *remain; // [Gpad]
    int8_t *winner, *err;  // [Gpad] written by K1
    C6GameHdr *hdr;        // [G]
    // root copy for the tree kernels
    int8_t *root_board;    // [G][cellsP]
    uint16_t *moves;       // [G][cells] cell of stone k
    uint16_t *hist_pi;     // [G][cells][cells] root child visits at move k
    // expanded-node records
    C6RecStat *rec_stat;   // [pages][64]
    float *rec_prior;      // [pages][64][cells]  n_prob of the child at each empty cell
    unsigned long long *page_free;   // [pages] bit i: record i of the page is free
    uint32_t *page_table;  // [G][maxpages] logical page -> pool page (C6_NOPAGE: none)
    C6HtEntry *ht;         // [chunks][1024]
    uint8_t *ht_tag;       // [chunks][1024] c6_ht_tag_of(key) of the slot's entry, 0 = empty
    uint32_t *chunk_table; // [G][maxchunks] logical chunk -> pool chunk
    uint32_t *page_ring, *chunk_ring;      // free ids
    unsigned long long *pool_ctr;          // [C6_POOL_N]
    C6RootChild *rootc;    // [G][cellsP] the root's children (valid while hdr.rootc_ok)
    // per-round leaf exchange
    int8_t *leaf_board;    // [G][cellsP]
    uint32_t *path_rec;    // [G][maxd] step i of the select path: record | chosen cell << 16 (maxd = cellsP: a path adds a stone per step)
    float *eval_prior;     // [G][cells]
    float *img_val;        // [G][8]
    float *img_pol;        // [G][8][cells]
    // move exchange
    int32_t *move_pos;     // [Gpad] -2: not moving, -1: no move (draw), else cell: pending for K1 / the next round's head
    int32_t *last_move;    // [Gpad] the same for the caller (c6_engine_last_moves): what the game played during the current step call
    int8_t *move_turn;     // [G]
    // finished games queue
    int *fin_winner, *fin_len;   // [finq_cap]
    uint32_t *fin_uid;
    int16_t *fin_moves;          // [finq_cap][cells]
    int finq_cap;
    // training tuples
    uint8_t *tuple_buf;
    unsigned long long tuple_cap;
    uint32_t *tuple_count;
    int tuple_stride;
    // constant tables
    const unsigned long long *zob; // [2][C6_MAXCELLS]
    const int16_t *aug_map;        // [8][cells] image k, cell c -> source cell of the board
    const int16_t *rec_map;        // [8][cells] image k, output cell c -> cell of image k's policy
    unsigned long long *counters;  // [C6_CNT_N] (rare events only: finished games, tuples, fetch queue)
};

// tree_kernels.hip
void c6_launch_search(const C6Dev &d, int kind, int nsims, hipStream_t s);      // built-in evaluators: nsims whole rounds (search + move) of every game per launch
void c6_launch_select(const C6Dev &d, hipStream_t s);
void c6_launch_expand(const C6Dev &d, hipStream_t s);
void c6_launch_recover(const C6Dev &d, hipStream_t s);
void c6_launch_augment(const C6Dev &d, float *img_boards, hipStream_t s);
void c6_launch_post_move(const C6Dev &d, hipStream_t s);
void c6_launch_reset_games(const C6Dev &d, int stagger, hipStream_t s);
// limit = tail of both pools' rings: launched in front of every kernel that may acquire pages (ids handed back by earlier launches
// become available)
// A launch helper that cannot run its kernel -- hipFuncSetAttribute refused an LDS request, a launch was rejected -- would leave the data
// of an earlier pass behind for whoever reads its outputs (the search would expand on stale priors).  The first such failure is kept
// (c6_note_hip) and every entry point that enqueues work -- c6_engine_step, c6_net_forward* -- ends with c6_launch_status(), which turns it
// into C6_ERR_HIP naming the call.
void c6_note_hip(hipError_t e, const char *what);
int c6_launch_status();      // C6_OK, or c6_set_error(C6_ERR_HIP, ...) for the failure noted since the last call (and forgets it)
void c6_launch_pool_snapshot(const C6Dev &d, hipStream_t s);
void c6_launch_ring_selftest(const uint32_t *ring, uint32_t mask, unsigned long long *ctr, uint32_t *count, uint32_t *popped, int waves, int tries, uint32_t seed, hipStream_t s);
// dynamic LDS of the kernels that keep a game's tag row / chunk list / page table per wave
size_t c6_tree_lds_bytes(const C6Dev &d);
// games in an error state -> the finished queue as they stand (winner 0, the stones played so far), agent_impl/mod.rs:163
void c6_launch_flush_errors(const C6Dev &d, hipStream_t s);
void c6_launch_noise_op(int N, int num, double alpha, double gd, double gc, const uint32_t *k0, const uint32_t *k1,
                        const uint32_t *search, const uint32_t *depth, const int8_t *boards, float *noise, hipStream_t s);
void c6_launch_augment_op(int N, int num, const int8_t *boards, const int16_t *aug_map, int8_t *out, hipStream_t s);
void c6_launch_recover_op(int N, int num, const int8_t *boards, const float *values8, const float *policies8,
                          const int16_t *rec_map, float *value, float *prior, hipStream_t s);

// board_kernel.hip  (K1)
void c6_launch_board_step(int N, int Gpad, int8_t *boards_soa, int8_t *turn, int8_t *remain, const int32_t *pos,
                          int8_t *winner, int8_t *err, hipStream_t s);
void c6_launch_search_winner(int N, int Gpad, const int8_t *boards_soa, int8_t *winner, hipStream_t s);
void c6_launch_legal_mask(int N, int Gpad, const int8_t *boards_soa, uint8_t *mask, hipStream_t s);

// net_kernels.hip  (K3 + K4)
// 3x3 convolution kernels (C6NetDev::variant), one per C6_PREC_* arithmetic:
enum {
    C6_CONV_F32N = 5,   // c6_conv3x3_f32n_kernel: FP32 direct (implicit GEMM on v_mfma_f32_32x32x2_f32), C6_PREC_FP32_DIRECT
    C6_CONV_B3S = 7,    // c6_conv3x3_b3s_kernel: split bf16 (3 bf16 MFMAs per product), C6_PREC_BF16X3
    C6_CONV_BF16 = 8,   // the same kernel with plain bf16 products: relaxed tolerance, C6_PREC_BF16
    C6_CONV_M8 = 9,     // c6_conv3x3_m8_kernel: fp16 main term + MX-fp8 cross terms, C6_PREC_F16F8
    C6_CONV_WINO = 10   // FP32 Winograd on v_mfma_f32_32x32x2_f32, C6_PREC_FP32 (default): c6_wino4_* (F(4x4,3x3), C = 256), c6_wino_* (F(2x2,3x3), C < 256)
};
struct C6NetDev {
    int N, cells, C, blocks;
    // packed weights (BatchNorm folded), see net_kernels.hip
    const float *stem_w;  // [9][2][C]
    const float *stem_b;  // [C]
    int variant;          // C6_CONV_*
    int wino6;            // C6_CONV_WINO at 19x19, C = 256, >= 2 residual blocks: conv_ww holds conv_wino6.hip's weight sets
    int num_cus;
    // per layer [(blocks-1)*2], only the image of the selected kernel is resident:
    const float *conv_ww;          // Winograd-transformed weights G g G^T packed per wave (conv_wino.hip), 16 C^2 floats
    const float *conv_w5;          // FP32 weights packed per wave (c6_conv3x3_f32n_kernel), 9 C^2 floats
    const unsigned short *conv_w7; // split bf16 hi/mid planes packed per wave (c6_conv3x3_b3s_kernel)
    const unsigned char *conv_w9;  // fp16 + fp8 planes packed per wave (c6_conv3x3_m8_kernel)
    const float *conv_b;  // [(blocks-1)*2][C]
    const float *pc_w;    // [2][C] folded
    const float *pc_b;    // [2]
    const float *pl_wT;   // [2*cells][cells]
    const float *pl_b;    // [cells]
    const float *vc_w;    // [C]
    const float *head_w;  // [3][C]: pc_w rows 0-1, vc_w -- the filters the last b3s / m8 convolution applies in its epilogue
    float vc_b;
    const float *vn_scale, *vn_shift; // [cells]
    const float *v1_wT;   // [cells][256]
    const float *v1_b;    // [256]
    const float *v2_w;    // [256]
    float v2_b;
    // range guard of the reduced-range arithmetic (C6_PREC_F16F8): device flag set by c6_range_check_kernel when an activation
    // leaves the range the fp16 / fp8 planes represent (nullptr: no check)
    unsigned int *range_flag;
    float range_limit;
};
// boards: int8 [nleaf][board_stride]; player: int8 of leaf i at player[i * player_stride]; images = 8 (dihedral, aug_map) or 1 (identity).
// act0/act1/act2: [nleaf*images][cells][C] f32 scratch; vbuf: c6_conv_wino_vbuf_floats(...) floats (C6_CONV_WINO only).
// Outputs pol [nleaf*images][cells], val [nleaf*images].
struct C6NetTimers;
void c6_net_forward_dev(const C6NetDev &net, int nleaf, int images, const int8_t *boards, int board_stride,
                        const int8_t *player, int player_stride, const int16_t *aug_map, float *act0, float *act1, float *act2, float *vbuf, float *ph,
                        float *vh, float *pol, float *val, hipStream_t s, C6NetTimers *timers);
void c6_conv_bench_launch(int variant, int C, int N, int nboards, const float *in, const float *w, const float *bias, const float *res,
                          float *out, float *vbuf, int num_cus, hipStream_t s);
// FP32 Winograd (conv_wino.hip: F(2x2,3x3) and the dispatch by channel count)
size_t c6_conv_packed_wino_floats(int C);
void c6_pack_conv_weights_wino(int C, const float *w_oihw_folded, float *packed);
size_t c6_conv_wino_vbuf_floats(int C, int N, int nboards);
bool c6_conv_wino_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                           float *vbuf, int num_cus, hipStream_t s, int relu = 1);
bool c6_conv_wino_transform(int C, int N, int nboards, const float *in, float *vbuf, int num_cus, hipStream_t s);      // the two steps of a dispatch
bool c6_conv_wino_gemm(int C, int N, int nboards, const float *vbuf, const float *wp, const float *bias, const float *res, float *out, int num_cus,
                       hipStream_t s, int relu = 1);
double c6_conv_wino_exec_flops(int C, int N, int nboards);
// the stem (dihedral gather + conv 2 -> C + folded BN + ReLU, c6_stem_kernel) as an INPUT of the Winograd kernels: the first layer's input
// transform computes it on the fly and the boundary kernel that needs it as a residual computes it again -- its activations are never stored
struct C6StemSrc {
    const int8_t *boards;
    int board_stride;
    const int8_t *player;
    int player_stride;
    const int16_t *aug_map;
    int images;
    const float *w, *bias;      // [9][2][C], [C]
};
// the heads' 1x1 filters applied by the tower's last output transform (conv_wino4.hip): w = [3][C] (policy rows 0-1, value), outputs ph / vh
// as c6_head_conv_kernel writes them; the layer's activations are then not stored
struct C6HeadFuse {
    const float *w, *pc_b;
    float vc_b;
    const float *vn_scale, *vn_shift;
    float *ph, *vh;
};
// returns 0: this form has no separate output transform; 1: done; 2: done and the heads' filters applied (heads != nullptr, F(4x4) only)
int c6_conv_wino_out(int C, int N, int nboards, float *vbuf, const float *bias, const float *res, float *out, int num_cus, hipStream_t s, int relu = 1,
                     const C6HeadFuse *heads = nullptr);
bool c6_conv_wino_mid(int C, int N, int nboards, float *vbuf, const float *bias, const float *res, float *act, int num_cus, hipStream_t s,
                      const C6StemSrc *res_stem = nullptr);
// F(4x4) only: the first layer's input transform straight from the boards (the stem computed on the fly; its activations stored only
// if `act` is given); false: not available
bool c6_conv_wino_stem_transform(int C, int N, int nboards, const C6StemSrc &stem, float *vbuf, float *act, int num_cus, hipStream_t s);
// FP32 Winograd F(4x4,3x3), C = 256 (conv_wino4.hip): what the functions above run at C = 256
size_t c6_conv_packed_wino4_floats(int C);
void c6_pack_conv_weights_wino4(int C, const float *w_oihw_folded, float *packed);
void c6_pack_conv_weights_wino4_dev(int C, const float *w_oihw_dev, float *packed_dev, int transpose_flip, hipStream_t s);
size_t c6_conv_wino4_vbuf_floats(int C, int N, int nboards);
bool c6_conv_wino4_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                            float *vbuf, int num_cus, hipStream_t s, int relu, int steps, const C6HeadFuse *heads = nullptr,
                            const C6StemSrc *stem = nullptr, float *zsep = nullptr,       // zsep: `vbuf` holds V only (c6_conv_wino4_v_floats), Z goes here
                            float *stat_partial = nullptr,       // r6: per-workgroup column sums of the outputs and their squares, [c6_conv_wino4_out_grid][2][C]
                            const float *const *bn_bwd = nullptr);      // ... or, with {aout, y, mean, var} of the BatchNorm + ReLU unit the outputs are the gradient OF: its backward sums
int c6_conv_wino4_out_grid(int N, int nboards, int num_cus);
size_t c6_conv_wino4_v_floats(int C, int N, int nboards);
double c6_conv_wino4_exec_flops(int C, int N, int nboards);
// weight gradient of the same convolution in the Winograd domain (trainer, FP32): dW (OIHW, overwritten) from dy and the layer's input x
size_t c6_conv_wino4_wgrad_partial_floats(int C, int N, int nboards, int num_cus);
bool c6_conv_wino4_wgrad(int C, int N, int nboards, const float *dy, const float *x, float *vbuf, float *partial, float *dw, int num_cus, hipStream_t s,
                         const float *vkept = nullptr);      // vkept: B^T x B of this layer as the forward pass left it (then x is not transformed again)
// the evaluator's tower at 19x19 and (r6) 15x15, C = 256, >= 2 residual blocks (conv_wino6.hip): the same F(4x4) Winograd pipeline on tiles six pixels wide
bool c6_w6_available(int C, int N, int blocks);
size_t c6_w6_packed_floats(void);
void c6_w6_pack_weights(const float *w_oihw_folded, float *packed);
size_t c6_w6_vbuf_floats(int N, int nboards);
// steps: 1 the first layer's input transform from the boards (`stem`), 2 the GEMMs, 16 the boundary kernel (bias, residual `res` or the
// recomputed stem, activations to `act` if given), 4 the last output transform with the heads' filters
bool c6_w6_dispatch(int N, int nboards, const float *wp, const float *bias, const float *res, float *act, float *vbuf, int num_cus, hipStream_t s, int steps,
                    const C6HeadFuse *heads = nullptr, const C6StemSrc *stem = nullptr);
double c6_w6_exec_flops(int N, int nboards);
// FP32 direct (conv_f32n.hip)
size_t c6_conv_packed_f32n_floats(int C);
void c6_pack_conv_weights_f32n(int C, const float *w_oihw_folded, float *packed);
bool c6_conv_f32n_dispatch(int C, int N, int nboards, const float *in, const float *wp, const float *bias, const float *res, float *out,
                           int num_cus, hipStream_t s, int relu = 1);
// split bf16 (conv_bf16x3.hip)
size_t c6_conv_packed_b3s_units(int C);
void c6_pack_conv_weights_b3s(int C, const float *w_oihw_folded, unsigned short *packed);
bool c6_conv_b3s_dispatch(int C, int N, int nboards, const float *in, const unsigned short *wp, const float *bias, const float *res,
                          float *out, int num_cus, hipStream_t s, int relu = 1, int split = 1, const float *head_w = nullptr);
int c6_conv_b3s_head_rows(int N);
// mixed fp16 + MX-fp8 products (conv_m8.hip, C6_PREC_F16F8)
size_t c6_conv_packed_m8_bytes(int C);
void c6_pack_conv_weights_m8(int C, const float *w_oihw_folded, unsigned char *packed);
// range_flag / range_hi: the arithmetic's range guard -- bit 0 of *range_flag is set when an output is not finite or beyond range_hi
bool c6_conv_m8_dispatch(int C, int N, int nboards, const float *in, const unsigned char *wp, const float *bias, const float *res,
                         float *out, int num_cus, hipStream_t s, int relu = 1, const float *head_w = nullptr,
                         unsigned int *range_flag = nullptr, float range_hi = 3.0e38f);
// the other half of the guard: the largest |activation| of a SAMPLE of `nboards` boards must reach range_lo (bit 1 of *range_flag
// otherwise; an all-zero layer is exact and passes): below it the fp8 cross-term planes run out of bits
void c6_launch_range_sample(const float *act, int nboards, size_t board_floats, float range_lo, float range_hi, unsigned int *range_flag, hipStream_t s);
// the validated range of C6_PREC_F16F8: per-board max |activation| of every tower layer in [2^-3, 3584] (3584: beyond it the fp8 plane of
// x_h / 8 saturates and the cross terms are clamped; 2^-3: the low parts x_l 2^8 drop below e4m3's normal range and the 16x margin of the
// parity tests, taken at activations of order 1, is used up), folded weights inside fp16 with every output channel within 2^-10 of the
// layer's largest (the fp8 weight planes share one shift per layer)
#define C6_M8_RANGE_HI 3584.0f
#define C6_M8_RANGE_LO 0.125f
#define C6_ACT_SLACK 1024   // floats of slack behind every activation buffer

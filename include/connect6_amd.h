/*
 * connect6_amd.h -- C ABI of libconnect6_amd.so, the MI355X-native self-play engine.
 *
 * Part 1 is the DROP-IN boundary: exactly the symbols, struct layouts and callback
 * conventions of the reference's C++ binding (staticlib `cppconnect6`), so that the
 * reference's single-header wrapper Connect6/cppbind/headers/connect6.hpp links
 * against this library unchanged.  Each declaration cites the reference interface it
 * replaces (paths relative to the reference checkout).
 *
 * Part 2 is the extended engine API (prefix c6_): run-time board size, thousands of
 * concurrent games resident in HBM, seeded C6RNG stream, resident policy/value net,
 * per-move pi / value outputs, steady-state stepping for benchmarks.  Plain pointers
 * and sizes only; no torch types.
 *
 * The library needs a gfx950 GPU for every compute entry point and fails loudly
 * (non-zero status + message from c6_last_error()) when none is present; there is
 * no CPU fallback.
 */
#ifndef CONNECT6_AMD_H
#define CONNECT6_AMD_H

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================== */
/* Part 1 -- drop-in boundary (N = 15, the reference's compiled BOARD_SIZE,  */
/*           Connect6/libconnect6/src/lib.rs:21, connect6.hpp:11)            */
/* ======================================================================== */

#define C6_REF_BOARD_SIZE 15
#define C6_REF_BOARD_CAPACITY (C6_REF_BOARD_SIZE * C6_REF_BOARD_SIZE)

/* cppbind/libcppconnect6/src/cppbind/cppeval/mod.rs:8-13, connect6.hpp:14.
 * On entry `policies` holds `len` boards as row-major f32 in {-1,0,+1} and `values`
 * is zero; the callee overwrites policies[len][N*N] and values[len].  The reference
 * always passes len == 8; this engine passes len == 8 * (games whose leaf has side
 * `player` to move), which a conforming callback (it loops over len) accepts. */
typedef void (*c6_callback)(int player, float *values, float *policies, int len);

/* cppbind/.../cpp_policy/mod.rs:8, connect6.hpp:15: board N*N f32, position preset to {-1,-1} */
typedef void (*c6_policy_callback)(float *board, int *position);

/* cppbind/.../rawobj/mod.rs:29-36 (RawPath), connect6.hpp:26-31.  sizeof == 912 */
typedef struct c6_path {
    int turn;
    int board[C6_REF_BOARD_SIZE][C6_REF_BOARD_SIZE];
    int row;
    int col;
} c6_path;

/* cppbind/.../rawobj/mod.rs:38-43 (RawPlayResult), connect6.hpp:33-37.  sizeof == 24 */
typedef struct c6_play_result {
    int winner;
    c6_path *paths;
    int len;
} c6_play_result;

/* cppbind/.../rawobj/mod.rs:45-50 (RawVec<RawPlayResult>), connect6.hpp:39-42.  sizeof == 16 */
typedef struct c6_vec {
    c6_play_result *vec;
    int len;
} c6_vec;

/* cppbind/.../rawobj/mod.rs:8 (AllocatorType<T>), connect6.hpp:17-23: caller-supplied `new T[n]` */
typedef c6_path *(*c6_alloc_path)(int size);
typedef c6_play_result *(*c6_alloc_result)(int size);
typedef int *(*c6_alloc_int)(int size);
typedef float *(*c6_alloc_float)(int size);

/* cppbind/libcppconnect6/src/lib.rs:20-58, connect6.hpp:44-48.
 * `num_game_thread` games driven by `callback`; stones are placed and winners found by the
 * HIP board kernel. */
c6_vec cpp_play(c6_policy_callback callback, c6_alloc_path alloc_path, c6_alloc_result alloc_result,
                bool debug, int num_game_thread);

/* cppbind/libcppconnect6/src/lib.rs:75-126, connect6.hpp:50-58.
 * `num_game_thread` concurrent AlphaZero self-play games on the GPU; leaves are evaluated by
 * `callback`, batched across games per side to move. */
c6_vec cpp_self_play(c6_callback callback, c6_alloc_path alloc_path, c6_alloc_result alloc_result,
                     int num_simulation, float epsilon, double dirichlet_alpha, float c_puct, bool debug,
                     int num_game_thread);

/* cppbind/libcppconnect6/src/lib.rs:140-169, connect6.hpp:60-65: MultiPolicy(black = AlphaZero over `callback`,
 * white = IoPolicy on stdin / stdout, policy/io_policy/mod.rs:70-107: "aB" = row a, column B; "invalid input, retry"),
 * Agent::debug(...).play(): the board is printed before every stone and each stone is logged as
 * "{player:?} ({row}, {col}), remain {n}, {s}.{ms} elapsed" (agent/agent_impl/mod.rs:121-178).  An occupied cell aborts
 * with "already set position" like the reference's unwrap(); end of input ends the game. */
c6_play_result cpp_play_with(c6_callback callback, c6_alloc_path alloc_path, int num_simulation, float epsilon,
                             double dirichlet_alpha, float c_puct);

/* FFI echo hooks, cppbind/.../ffi_test/mod.rs:13-224, connect6.hpp:68-93 (host only, no GPU) */
typedef struct c6_vec_int { int *vec; int len; } c6_vec_int;
typedef struct c6_vec_float { float *vec; int len; } c6_vec_float;
c6_path test_new_raw_path(void);                                                  /* ffi_test/mod.rs:13 */
c6_path test_with_raw_path(void);                                                 /* :19 */
c6_path test_echo_raw_path(int turn, int *board, int row, int col);               /* :38 */
c6_play_result test_with_raw_play_result(c6_alloc_path allocator);                /* :61 */
c6_play_result test_echo_raw_play_result(int winner, c6_path *path, int len, c6_alloc_path allocator); /* :90 */
c6_vec_int test_with_raw_vec(c6_alloc_int allocator);                             /* :125 */
c6_vec_int test_echo_raw_vec(int *ptr, int len, c6_alloc_int allocator);          /* :133 */
c6_vec_float test_echo_cppeval(int turn, int *boards, int len, c6_callback callback, c6_alloc_float allocator); /* :149 */
c6_vec_int test_cpp_policy(float *board, c6_policy_callback callback, c6_alloc_int allocator);                /* :201 */

/* ======================================================================== */
/* Part 2 -- extended engine API                                             */
/* ======================================================================== */

enum {
    C6_OK = 0,
    C6_ERR_NO_GPU = 1,        /* no gfx950 device / HIP runtime error */
    C6_ERR_BAD_ARG = 2,
    C6_ERR_OVERFLOW = 3,      /* a per-game record pool or path buffer overflowed */
    C6_ERR_HIP = 4,
    C6_ERR_INVALID_POSITION = 5, /* game/game_impl/mod.rs:151-153 "invalid position" */
    C6_ERR_ALREADY_SET = 6,      /* game/game_impl/mod.rs:155-157 "already set position" */
    C6_ERR_RANGE = 7          /* C6_PREC_F16F8 only: weights (at load) or activations (while playing) left the range that arithmetic
                                 is validated for -- the net's outputs would no longer be the FP32 net's.  Use C6_PREC_FP32 / _BF16X3 */
};

/* evaluator kinds (policy/alphazero_policy/mod.rs:128-134 `Evaluator`) */
enum {
    C6_EVAL_RANDOM = 1,   /* RandomEvaluator (mod.rs:137-169) on the seeded C6RNG stream, on device */
    C6_EVAL_UNIFORM = 2,  /* value 0, policy 1/N^2 on device ("zero-cost evaluator", BASELINE config 5) */
    C6_EVAL_NET = 3,      /* resident policy/value ResNet (AlphaZero/cpp_weighted/model.hpp:9-95) */
    C6_EVAL_CALLBACK = 4, /* host callback with the reference's signature */
    C6_EVAL_STRESS = 5    /* soak test of the record pools, on device: games whose uid is a multiple of `stress_every` get a prior peaked on
                             the innermost ring of empty cells around the board centre -- prior(c) = 2^(-20 min(d2(c) - d2min, 6)), d2 = squared
                             distance to the centre, d2min its minimum over the empty cells -- and value -0.875 * side to move (with a small
                             epsilon nearly every search follows the peak level after level: the tree grows DEEP and keeps several times
                             num_simulation records alive); the other games get the uniform evaluator.  All values are powers of two /
                             multiples of 1/8: the dihedral average is exact and a CPU evaluator reproduces it bit for bit */
};

/* Arithmetic of the net's 3x3 convolutions (everything else is FP32).  The reference net is f32 (AlphaZero/cpp_weighted/model.hpp:52-95),
 * and so is the default. */
enum {
    C6_PREC_DEFAULT = 0,     /* = C6_PREC_FP32 */
    C6_PREC_FP32 = 1,        /* f32 operands, exact f32 products, f32 accumulate on v_mfma_f32_32x32x2_f32, evaluated in Winograd form: at 256
                                channels tiles of F(6,3) / F(4,3) / F(3,3) across and F(4,3) / F(3,3) down a 19x19 or 15x15 board (4.1x fewer matrix
                                products than the direct form), F(4x4,3x3) at 9x9 and in two-block towers, F(2x2,3x3) at 64 / 128 channels (2.03x):
                                the same arithmetic type, the whole FP32 range */
    C6_PREC_BF16X3 = 3,      /* operands split hi+mid in bf16, x_hi*w_hi + x_hi*w_mid + x_mid*w_hi on bf16 MFMAs, f32 accumulate: error
                                <= 3*2^-16 per product; covers the whole FP32 range.  Net outputs within 1e-5 of FP32 in the tests */
    C6_PREC_BF16 = 4,        /* plain bf16 products (x_hi*w_hi only): RELAXED tolerance -- net outputs agree with FP32 to ~1e-3, NOT the 1e-5
                                parity grade; for evaluation matches / throughput headroom.  Never the benchmark's `value` */
    C6_PREC_F16F8 = 5,       /* opt-in.  x_h*w_h on fp16 MFMAs + the cross terms x_h*w_l + x_l*w_h on MX-scaled fp8 MFMAs, f32 accumulate:
                                ~2^-15 per product like C6_PREC_BF16X3 at 70 % of its matrix-pipe cycles, but a BOUNDED range: per-board
                                max |activation| of every tower layer in [2^-3, 3584], BatchNorm-folded weights inside fp16 with the output
                                channels of a layer within 2^10 of each other.  Guarded: c6_engine_load_weights returns C6_ERR_RANGE for weights
                                outside it, and activations that leave it while playing set a sticky device flag that c6_engine_run /
                                _stats / _maintain / _fetch_* report as C6_ERR_RANGE -- wrong priors are never returned with C6_OK */
    C6_PREC_FP32_DIRECT = 6  /* the same f32 arithmetic as C6_PREC_FP32 as a direct implicit GEMM (no Winograd transforms): round 1's FP32
                                kernel, kept as the yardstick of the Winograd path */
};

/* policy/alphazero_policy/mod.rs:91-96 HyperParameter + engine geometry */
typedef struct c6_config {
    int board_size;          /* 9, 15 or 19 */
    int num_games;           /* concurrent games on this device */
    int num_simulation;      /* HyperParameter.num_simulation */
    float epsilon;           /* HyperParameter.epsilon */
    double dirichlet_alpha;  /* HyperParameter.dirichlet_alpha */
    float c_puct;            /* HyperParameter.c_puct */
    uint64_t seed;           /* C6RNG seed */
    uint32_t game_uid_base;  /* game slot i starts with uid = base + i (C6RNG key) */
    uint32_t game_uid_stride;/* a recycled slot continues with uid += stride (0: num_games) */
    int evaluator;           /* C6_EVAL_* */
    int net_channels;        /* C (reference: 256) */
    int net_blocks;          /* reference `num_block`: stem + (num_block-1) residual blocks */
    int records_per_game;    /* AVERAGE expanded-node records per game the shared pool is sized for at creation (0: 4 * num_simulation + 64;
                                whole pages of 64, never less than one move's searches + a page).  The pool is shared: a game holds what
                                its tree needs, and more memory is mapped when the pool runs low (c6_engine_run / c6_engine_maintain) */
    int recycle;             /* 1: a finished game slot restarts at once (steady-state mode) */
    int stagger;             /* 1: game i starts its first move with (i mod num_simulation) simulations
                                already counted, so every round advances num_games/num_simulation games */
    int device;              /* HIP device ordinal */
    int net_chunk_boards;    /* 0: automatic; boards (dihedral images) per net pass */
    int net_precision;       /* C6_PREC_*: arithmetic of the 3x3 convolutions (everything else is FP32) */
    int max_records_per_game;/* 0: 65000 (record indices travel as 16 bits).  The most records ONE game may hold */
    int stress_every;        /* C6_EVAL_STRESS: games with uid % stress_every == 0 get the peaked evaluator (0: every game) */
} c6_config;

typedef struct c6_engine c6_engine;

const char *c6_last_error(void);
const char *c6_version(void);
/* the random-stream specification the engine's noise / tie-breaks / built-in evaluators follow ("C6RNG v5": DESIGN.md 3).  The games an
 * engine plays are a function of (this, seed, weights, arithmetic, hyper-parameters): records of played games carry it. */
const char *c6_rng_spec(void);
int c6_device_count(void);

void c6_default_config(c6_config *cfg); /* HyperParameter::default (mod.rs:98-108), N = 15 */
int c6_engine_create(const c6_config *cfg, c6_engine **out);
void c6_engine_destroy(c6_engine *e);

/* Weight blob (host pointer): flat f32, the tensors of WeightedPolicy (AlphaZero/cpp_weighted/model.hpp:10-50) in FORWARD order, each
 * BatchNorm's running statistics right behind its affine parameters, every tensor in PyTorch's layout:
 *   conv.weight [C,2,3,3], conv.bias, norm.weight, norm.bias, norm.running_mean, norm.running_var;
 *   for each residual block i, for j = 0, 1: conv{i}_{j}.weight [C,C,3,3], .bias, norm{i}_{j}.weight, .bias, .running_mean, .running_var;
 *   policy_conv.weight [2,C,1,1], .bias, policy_norm.{weight, bias, running_mean, running_var}, policy_linear.weight [N^2,2N^2], .bias;
 *   value_conv.weight [1,C,1,1], .bias, value_norm.{weight, bias, running_mean, running_var} [N^2], value_linear1.weight [256,N^2], .bias,
 *   value_linear2.weight [1,256], .bias.
 * This is NOT libtorch's registration order (model.hpp registers the heads before the tower and lists buffers separately): build the
 * blob by tensor NAME -- alphazero_connect6_amd.blob_from_state_dict does, INTEGRATION.md has the C++ loop.  A blob may carry the
 * 8-float header of c6_net_blob_write_header in front (magic, version, geometry, payload size, checksum): c6_engine_load_weights
 * accepts both sizes and rejects a headed blob whose geometry or checksum does not match. */
size_t c6_net_blob_floats(int board_size, int channels, int blocks);
#define C6_BLOB_HEADER_FLOATS 8
size_t c6_net_blob_header_floats(void);
/* dst = header + payload (payload already in place behind the header); nfloats = c6_net_blob_floats(...) + C6_BLOB_HEADER_FLOATS */
int c6_net_blob_write_header(float *dst, size_t nfloats, int board_size, int channels, int blocks);
int c6_net_blob_check_header(const float *blob, size_t nfloats, int board_size, int channels, int blocks);
/* C6_ERR_RANGE: the engine's net_precision is C6_PREC_F16F8 and the weights are outside its validated range (message says which layer) */
int c6_engine_load_weights(c6_engine *e, const float *blob, size_t nfloats);
int c6_engine_set_callback(c6_engine *e, c6_callback cb);

/* Run `rounds` simulation rounds: every live game does one search() (mod.rs:411-426) per round;
 * a game whose simulation count reaches num_simulation makes its move (mod.rs:464-489 +
 * agent/agent_impl/mod.rs:131-171).  Asynchronous on the engine's stream unless noted. */
int c6_engine_step(c6_engine *e, int rounds);
/* step until every game has finished (recycle == 0) or max_rounds; synchronous.  The reference's node map grows without limit
 * (alphazero_policy/mod.rs:186-190); here the per-game record pool is doubled whenever a game runs out or the pools pass 3/4
 * (a stopped game resumes exactly where it was).  If that is impossible (65 000 records per game, or device memory), or a game
 * produced NaN scores, the call returns C6_ERR_OVERFLOW -- after all other games have finished -- and those games are in the
 * finished queue as PARTIAL results: winner 0, the stones played so far. */
int c6_engine_run(c6_engine *e, long max_rounds);
/* c6_engine_step never synchronises, so it cannot grow the pools: a game that runs out there waits in status overflow (in
 * recycle mode its slot is abandoned at the next move round: c6_stats.abandoned_games).  Callers that drive the engine with
 * c6_engine_step call this every now and then (it synchronises): grows the pools when a game is waiting or they pass 3/4. */
int c6_engine_maintain(c6_engine *e);
int c6_engine_sync(c6_engine *e);

/* Evaluation games MultiPolicy(this engine's AlphaZero search, opponent) for all games at once, on the device from the first stone to
 * the last (policy/multi_policy/mod.rs:61-71: one policy per colour).  engine_colour: -1 the engine plays Black, +1 White.  opponent:
 * C6_OPP_RANDOM = RandomPolicy (policy/random_policy/mod.rs:39-47; its unseeded pick becomes C6RNG tag 8: word 0 of Philox(key =
 * (seed_lo, seed_hi + game uid); ctr = (stones on the board, 0, 0, 8)), the k-th empty cell in row-major order, k = (w * len) >> 32);
 * C6_OPP_ROLLOUT = DefaultPolicy, pure MCTS with random playouts: `r` from c6_rollout_create with num_games slots (part 5).
 * The engine must have been created with recycle = stagger = 0 and be at the start of its games; results: c6_engine_fetch_games.
 * max_stones <= 0: to the end.  No board, move or result crosses the host while the games are played. */
enum { C6_OPP_RANDOM = 1, C6_OPP_ROLLOUT = 2 };
struct c6_rollout;
int c6_engine_play_match(c6_engine *e, int engine_colour, int opponent, struct c6_rollout *r, uint64_t opp_seed, long max_stones);

/* the canonical game state (what K1 reads and writes): boards[g][N*N] in {-1, 0, +1}, the side to move and the stones it still has
 * to place (Game{turn, num_remain}, game_impl/mod.rs:117-121); any of the pointers may be NULL.  Synchronises. */
int c6_engine_boards(c6_engine *e, int8_t *boards, int8_t *turn, int8_t *remain);

typedef struct c6_stats {
    uint64_t rounds;          /* rounds stepped */
    uint64_t simulations;     /* searches performed (all games) */
    uint64_t select_steps;    /* PUCT evaluations (sum of select depths) */
    uint64_t stones;          /* stones placed */
    uint64_t games_finished;
    uint64_t tuples;          /* training tuples emitted */
    uint32_t live_games;
    uint32_t overflow_games;
    uint32_t max_records_used;
    uint32_t max_depth;
    uint64_t abandoned_games;  /* games given up in an error state (recycled slots, or flushed as partial results by c6_engine_run) */
    uint32_t records_per_game; /* records ONE game may hold right now without the host (64 x the per-game page cap; raised by
                                  c6_engine_maintain up to max_records_per_game) */
    uint32_t pool_grows;       /* times memory was mapped into the pools after creation */
    uint32_t pool_pages;       /* record pages (64 records each) in the pool shared by all games */
    uint32_t pool_pages_free;  /* ... of which no game holds now */
    uint32_t pool_pages_low;   /* fewest free pages any acquisition has seen */
    uint32_t cap_raises;       /* times the per-game caps were raised */
    uint32_t parked_games;     /* games waiting at a move boundary for pages / chunks / caps (they are part of overflow_games) */
    uint32_t max_pages_held;   /* most pages one game holds now */
    uint64_t pool_bytes;       /* device memory mapped for the two pools */
} c6_stats;
int c6_engine_stats(c6_engine *e, c6_stats *out); /* synchronises */

/* Invariants of the two pools (a self-check for soak tests): every record page / table chunk is held by exactly one game or sits
 * exactly once in its free ring.  *_dup: ids seen more than once (two owners, or an owner and the ring), *_leaked: ids seen nowhere,
 * *_bad: ids beyond the pool.  All three are 0 on a sound engine, whatever was parked, abandoned or recycled.  Synchronises. */
typedef struct c6_pool_audit {
    uint32_t pages_total, pages_held, pages_free, pages_dup, pages_leaked, pages_bad;
    uint32_t chunks_total, chunks_held, chunks_free, chunks_dup, chunks_leaked, chunks_bad;
} c6_pool_audit;
int c6_engine_pool_audit(c6_engine *e, c6_pool_audit *out);

/* finished games (winner, moves): agent::PlayResult, agent/agent_impl/mod.rs:41-44.
 * Copies up to `cap` finished, not yet fetched games: winners[i], lens[i], uids[i],
 * moves[i*N*N + k] = cell of stone k.  Returns the number copied in *count; a caller with room for fewer games than are
 * queued gets the next `cap` of them at every call (0: none left).  The device keeps at most max(4 G, 1024) results between
 * two full drains (steady-state mode: fetch at least that often; an overflow is said on stderr, c6_stats.games_finished counts all). */
int c6_engine_fetch_games(c6_engine *e, int cap, int *count, int *winners, int *lens, uint32_t *uids,
                          int16_t *moves);

/* training tuples (state, pi, turn, z): device buffer of fixed-size records
 * [state int8 N*N | pad to even | pi u16 N*N (root child visit counts, mod.rs:433-452) | ... | pos u16 | turn int8 | z int8]
 * (pos = row*N+col of the stone played, agent::Path.pos; the last four bytes of a record are pos, turn, z),
 * padded to a multiple of 4 bytes.  The caller may supply the buffer (e.g. torch-owned memory
 * handed to RCCL all-gather); count_dev is a device uint32. */
size_t c6_tuple_stride(int board_size);
int c6_engine_set_tuple_buffer(c6_engine *e, void *dev_buf, size_t capacity_records, uint32_t *count_dev);
int c6_engine_enable_tuples(c6_engine *e, size_t capacity_records); /* engine-owned tuple buffer */
/* The same as c6_engine_set_tuple_buffer WITHOUT draining the engine's stream: rounds enqueued before the call keep writing the
 * old buffer, rounds enqueued after it write the new one (the buffer travels with each launch).  With c6_engine_stream this is
 * what a double-buffered exchange needs (alphazero-connect6_amd/shard.py TuplePool: the retired buffer is all-gathered on a
 * side stream while the engine plays on) -- the caller orders the streams with events. */
int c6_engine_swap_tuple_buffer(c6_engine *e, void *dev_buf, size_t capacity_records, uint32_t *count_dev);
/* the HIP stream (hipStream_t as void*) every engine launch goes to: record / wait events on it to order other streams against
 * the engine's rounds without a host synchronisation */
void *c6_engine_stream(c6_engine *e);
/* copies up to `cap` records to host_buf and removes them from the buffer; records beyond `cap` stay for the next call (host_buf NULL: discards all) */
int c6_engine_fetch_tuples(c6_engine *e, size_t cap, size_t *count, void *host_buf);

/* inspection for parity tests: the root of game `g`, synchronises.
 * visits[c] = child visit (-1 where the cell is occupied), q[c] = child q_value, prior[c] = child n_prob */
typedef struct c6_root_info {
    int turn, remain, stones, status, winner;
    int root_visit;
    float root_value, root_q_sum, root_q_value;
    uint32_t nsearch;
    int records_used;
} c6_root_info;
int c6_engine_root(c6_engine *e, int g, c6_root_info *info, int8_t *board, int32_t *visits, float *q, float *prior);

/* stones from outside the engine's policy: MultiPolicy (policy/multi_policy/mod.rs:61-71: another policy plays one
 * colour) and the human side of cpp_play_with (lib.rs:140-169).  pos[g] = cell (row*N + col) or -2 = none for game g.
 * Runs Game::set / is_game_end and re-roots the engine's tree below the new position (tree reuse as in
 * AlphaZero::next, alphazero_policy/mod.rs:464-489).  An occupied or out-of-range cell makes the game's status an error
 * (the reference: Err -> unwrap() panic). */
int c6_engine_apply_moves(c6_engine *e, const int32_t *pos);
/* the cells the engine's policy played during the last c6_engine_step call: pos[g] = cell, -1 = no move (draw), -2 = game g did not
 * move in that call (a game moves at most once per num_simulation rounds).  Two
 * engines with different weights, each searching for one colour and receiving the other's stones through
 * c6_engine_apply_moves, are MultiPolicy for resident nets: net-vs-net evaluation games (tools/net_vs_net.py). */
int c6_engine_last_moves(c6_engine *e, int32_t *pos);

/* timing of the dominant kernel (3x3 conv) for the roofline: enables HIP-event timing on the
 * engine's stream; totals since the last reset. */
int c6_engine_profile(c6_engine *e, int enable);
typedef struct c6_profile {
    double conv_ms;        /* summed duration of the conv3x3 layers (all kernels of a layer) */
    uint64_t conv_launches; /* layers timed */
    double conv_flops;     /* algorithmic flops of those launches */
    double select_ms, expand_ms, eval_other_ms, move_ms;
    uint64_t select_launches, expand_launches;
    double conv_pre_ms;     /* the part of conv_ms spent in the Winograd input transform (C6_PREC_FP32), 0 otherwise */
    double conv_exec_flops; /* matrix FLOPs those launches EXECUTED (padding rows / tiles included; Winograd: 16 or 36 C^2 per tile) */
    double conv_post_ms;    /* the part of conv_ms spent in the Winograd F(4x4) output transform (C6_PREC_FP32 at 256 channels), 0 otherwise */
} c6_profile;
int c6_engine_profile_read(c6_engine *e, c6_profile *out, int reset);

/* ---- stand-alone batched operators (host buffers in/out; used by parity tests) ---- */

/* K1 board kernel: boards SoA int8 [N*N][G] (cell-major, game minor), turn/remain int8 [G],
 * pos int32 [G] = row*N+col, or <0 for "no move" ("row >= N" etc. must be passed as pos >= N*N).
 * Game::set + is_game_end (game_impl/mod.rs:148-169,249-252).  Outputs winner int8 [G] and
 * err int8 [G] (C6_OK / C6_ERR_INVALID_POSITION / C6_ERR_ALREADY_SET). */
int c6_board_step(int board_size, int num_games, int8_t *boards_soa, int8_t *turn, int8_t *remain,
                  const int32_t *pos, int8_t *winner, int8_t *err);
/* search (search_winner/mod.rs:201-246) on `num` boards, SoA layout as above */
int c6_search_winner(int board_size, int num, const int8_t *boards_soa, int8_t *winner);
/* legal-move bitmask: mask[(c>>3)*G + g] bit (c&7) set iff cell c of game g is empty */
int c6_legal_mask(int board_size, int num, const int8_t *boards_soa, uint8_t *mask);
/* augment_way8 (augment.rs:60-73): boards int8 [num][N*N] -> out [num][8][N*N] */
int c6_augment8(int board_size, int num, const int8_t *boards, int8_t *out);
/* get_from tail (mod.rs:222-233): value = sum(values)/8, recover_way8, mask occupied */
int c6_recover8(int board_size, int num, const int8_t *boards, const float *values8, const float *policies8,
                float *value, float *prior);
/* net forward (model.hpp:52-95) on `num` boards (int8, no augmentation) with per-board player */
int c6_net_forward(int board_size, int channels, int blocks, const float *blob, size_t nfloats, int num,
                   const int8_t *boards, const int8_t *player, float *policy, float *value);
/* same with an explicit C6_PREC_* */
int c6_net_forward_prec(int board_size, int channels, int blocks, int precision, const float *blob, size_t nfloats, int num,
                        const int8_t *boards, const int8_t *player, float *policy, float *value);
/* development / profiling hook: average duration (ms) of `iters` launches of the 3x3 convolution of one C6_PREC_* on random data
 * (`ablate` is unused and must be 0) */
int c6_bench_conv(int board_size, int channels, int nboards, int iters, int precision, int ablate, int with_residual, double *avg_ms);
/* host-only (no GPU needed): the weight image c6_conv3x3_m8_kernel (C6_PREC_F16F8) reads for one layer's folded OIHW weights --
 * fp16 planes [slice][tap 0..9 (9 = zeros)][wave][16-channel tile][lane][8], then per tap pair the fp8 planes
 * [slice][pair][wave][tile][lane][16 x q8(w_l 2^16) | 16 x q8(w_h 2^5)] (OCP e4m3, round to nearest even, saturating) */
size_t c6_conv_pack_m8_bytes(int channels);
int c6_conv_pack_m8(int channels, const float *w_oihw, unsigned char *packed, size_t nbytes);
/* K1 alone: `num_games` resident games, one stone per game per launch (average ms over `iters` launches) */
int c6_bench_board(int board_size, int num_games, int iters, double *avg_ms);
/* K1's search-only launch (is_game_end of every resident game, no stone: N*N bytes read + 1 written per board) */
int c6_bench_board_scan(int board_size, int num_games, int iters, double *avg_ms);
/* Self-test of the pools' free-id rings (the pop protocol of tree_kernels.hip on its own): `launches` launches of `waves` waves, each
 * asking `tries` times for 1-8 ids from a ring of `ids` ids, so that nearly all of every launch runs on a ring that is almost dry.
 * *dup: ids handed out more than once; *over: ids handed out beyond the ring's content; *min_popped: fewest ids one launch handed out
 * (a sound protocol gives dup = over = 0 and hands out nearly all: failures in flight may leave at most a few behind). */
int c6_pool_ring_selftest(int ids, int waves, int tries, int launches, uint32_t seed, uint32_t *dup, uint32_t *over, uint32_t *min_popped);
/* C6RNG Dirichlet noise of select step (search, depth) for `num` boards (keys k0/k1 per board) */
int c6_dirichlet_noise(int board_size, int num, double alpha, const uint32_t *k0, const uint32_t *k1,
                       const uint32_t *search, const uint32_t *depth, const int8_t *boards, float *noise);

/* ---- replay buffer: device-resident restatement of AlphaZero/cpp_weighted/ReplayBuffer.hpp ---- */
typedef struct c6_replay c6_replay;
int c6_replay_create(int board_size, size_t max_size, c6_replay **out);      /* ReplayBuffer(max_size, .), :12-20; on the calling thread's current device */
int c6_replay_create_on_device(int board_size, size_t max_size, int device, c6_replay **out);   /* ... on a named one (e.g. a c6_group member's) */
void c6_replay_destroy(c6_replay *r);
/* push_game (:22-29) for every record of a tuple buffer.  _dev: records and their count live on the device (e.g. the
 * all-gathered tuple buffer), count clamped to `cap`; no host synchronisation. */
int c6_replay_push_dev(c6_replay *r, const void *tuples_dev, const uint32_t *count_dev, size_t cap);
int c6_replay_push_host(c6_replay *r, const void *tuples_host, size_t count);
/* everything the engine has emitted since the last fetch/push goes device-to-device into the ring; the engine's tuple
 * counter is reset */
int c6_engine_push_tuples(c6_engine *e, c6_replay *r);
int c6_replay_clear(c6_replay *r);                                           /* :31-33 */
int c6_replay_size(c6_replay *r, size_t *num_data, size_t *tail);             /* :35-37 */
/* sample (:39-65): num_sample draws WITH replacement, index_i = C6RNG(seed; i, draw_id) % num_data.  Host outputs:
 * winners[n], players[n], boards[n*N*N] (f32 -1/0/1), poses[n] (row*N+col) as the reference; pi[n*N*N] optional
 * (visit counts as f32, build-side extension). */
int c6_replay_sample(c6_replay *r, int num_sample, uint64_t seed, uint32_t draw_id, float *winners, float *players,
                     float *boards, float *poses, float *pi);

/* the same draw into DEVICE buffers on a caller's HIP stream (hipStream_t passed as void*): the trainer's mini-batch */
int c6_replay_sample_device(c6_replay *r, int num_sample, uint64_t seed, uint32_t draw_id, float *winners, float *players,
                            float *boards, float *poses, void *stream);
int c6_replay_board_size(const c6_replay *r);

/* =====================================================================================================================
 * Part 4 -- training step: the body of the reference's train loop,
 *     optimizer.zero_grad(); loss = model->loss(winners, players, boards, poses); loss.backward(); optimizer.step();
 * (AlphaZero/cpp_weighted/main.cpp:151-157) with model = WeightedPolicy in train() mode (model.hpp:52-108: BatchNorm on
 * batch statistics, running statistics updated with momentum 0.1; loss = mean((winner - value)^2) with the reference's
 * [B] - [B,1] broadcast + mean(-log(policy[pos] + 1e-7))) and torch::optim::SGD(lr).momentum(m) (main.cpp:107-109).
 * Weights use the blob of c6_engine_load_weights, so c6_trainer_export_weights -> c6_engine_load_weights closes the
 * self-play / train loop.  FP32; every reduction has a fixed order (a step is bit-reproducible).
 * ===================================================================================================================== */
typedef struct c6_trainer c6_trainer;

/* batch = mini_batch (ReplayBuffer::num_sample, main.cpp:111); lr / momentum as torch::optim::SGDOptions */
int c6_trainer_create(int board_size, int channels, int blocks, int batch, float lr, float momentum, int device, c6_trainer **out);
void c6_trainer_destroy(c6_trainer *t);
int c6_trainer_set_lr(c6_trainer *t, float lr, float momentum);
/* precision of the 3x3 convolutions (forward pass, input gradient, weight gradient).  C6_PREC_DEFAULT = C6_PREC_FP32 -- the same meaning
 * it has in the engine, and the trainer's initial setting: f32 like the reference trainer (main.cpp:107-157), at 256 channels in the
 * Winograd domain; gradient parity with an FP32 framework to ~1e-6.  C6_PREC_FP32_DIRECT: the same arithmetic on the direct kernels
 * (the yardstick).  C6_PREC_BF16X3: the evaluator's split-bf16 kernels, weight gradient included -- 1.3x faster, gradients to ~1e-3. */
int c6_trainer_set_precision(c6_trainer *t, int precision);
/* parity-test operator: dW (OIHW [C][C][3][3], f32) of y = conv3x3(x, w) from dy and x (host buffers, NHWC [B][N*N][C]).
 * method 0: the direct kernel, 1: in the Winograd domain (C = 256) -- the two implementations the FP32 trainer chooses between */
int c6_conv_wgrad_op(int n, int C, int B, const float *dy, const float *x, int method, float *dw);
/* torch::load / torch::save (main.cpp:100,163) in blob form; load resets the momentum buffers */
int c6_trainer_load_weights(c6_trainer *t, const float *blob, size_t nfloats);
int c6_trainer_export_weights(c6_trainer *t, float *blob, size_t nfloats);
/* one optimisation step on the batch (winners[B], players[B], boards[B*N*N], poses[B] as ReplayBuffer::sample returns
 * them; on_device != 0: the pointers are device memory).  *loss_out = the loss before the step (may be NULL). */
int c6_trainer_step(c6_trainer *t, const float *winners, const float *players, const float *boards, const float *poses, int on_device,
                    float *loss_out);
/* model->loss(...) alone, in train mode like main.cpp:159-160 (running statistics move, parameters do not) */
int c6_trainer_loss(c6_trainer *t, const float *winners, const float *players, const float *boards, const float *poses, int on_device,
                    float *loss_out);
/* buffer.sample() + step on the device-resident replay buffer, no host round trip */
int c6_trainer_step_replay(c6_trainer *t, c6_replay *r, uint64_t seed, uint32_t draw_id, float *loss_out);
/* gradients of the last step in blob layout (zeros at the BatchNorm running statistics) -- for the parity tests */
int c6_trainer_gradients(c6_trainer *t, float *blob, size_t nfloats);
const char *c6_trainer_last_error(void);

/* =====================================================================================================================
 * Part 5 -- DefaultPolicy, pure MCTS with random playouts: policy/default_policy/mod.rs:84-275 for
 * `num_games` independent policies at once, one GPU thread per policy, trees resident in HBM.  c6_rollout_next is
 * Policy::next (:262-273): num_iter x search (:213-236: select :119-148, expand :150-170, playout + update :172-211), then
 * policy (:238-258).  The reference's quirks are kept (the map is never pruned, max_by takes the last of equal children,
 * White stops at a child Black never won, expand resets a node that already exists and pushes duplicates); its
 * `thread_rng` picks become the k-th empty cell in row-major order with k = (w * len) >> 32, w = the next word of
 * C6RNG Philox(key = (seed_lo, seed_hi + uid); ctr = (draw >> 2, 0, search, 6)), the fallback of `policy` word 0 of
 * ctr = (0, 0, searches so far, 7); uid = slot + num_games * (resets of the slot).  The board key is a 64-bit Zobrist hash
 * (the reference: 64-bit SipHash).  tests/test_gpu_rollout.py holds it bit-exact to oracle/c6_default.c.
 * Turn and stones-left are derived from the stone count (Black 1, then 2 each: game/game_impl/mod.rs:125-169).
 * ===================================================================================================================== */
typedef struct c6_rollout c6_rollout;

/* num_iter: DefaultPolicy::with_num_iter (:103-108; new() = 50).  max_nodes = nodes (and edges) per tree, 0 = enough for a
 * whole game with both sides on this policy (num_iter * N*N + N*N + 1).  C6_ERR_OVERFLOW from next() when a tree fills up. */
int c6_rollout_create(int board_size, int num_games, int num_iter, uint64_t seed, size_t max_nodes, int device, c6_rollout **out);
void c6_rollout_destroy(c6_rollout *r);
/* DefaultPolicy::new for the slots with mask[g] != 0 (all when NULL): an empty map and the next uid of the slot */
int c6_rollout_reset(c6_rollout *r, const uint8_t *mask);
/* boards int8 [num_games][N*N] (-1 Black / 0 / 1 White), active uint8 [num_games] or NULL = all, pos int32 [num_games] out:
 * row * N + col, or -1 (inactive slot / no empty cell).  on_device != 0: the three pointers are device memory. */
int c6_rollout_next(c6_rollout *r, const int8_t *boards, const uint8_t *active, int32_t *pos, int on_device);
typedef struct c6_rollout_info {
    uint32_t nodes, edges;   /* map.len() and the sum of next_node.len() ever pushed */
    uint32_t searches;       /* search() calls since the reset */
    uint32_t uid;
} c6_rollout_info;
int c6_rollout_slot_info(c6_rollout *r, int g, c6_rollout_info *out);
/* the Node of `board` in slot g (test access): *found, its visit / black_win, and up to cap children in push order
 * (cell, visit, black_win); *n_children = next_node.len() */
int c6_rollout_node(c6_rollout *r, int g, const int8_t *board, int *found, int *visit, int *black_win, int *n_children, int cap,
                    int32_t *child_cell, int32_t *child_visit, int32_t *child_black_win);
/* playouts only, for the bench: `iters` x (uniformly random playout from every board to the end); *avg_ms per launch,
 * *steps = stones placed per launch, winners int8 [num_games] of the last launch (may be NULL) */
int c6_rollout_bench_playouts(c6_rollout *r, const int8_t *boards, int iters, double *avg_ms, double *steps, int8_t *winners);

/* =====================================================================================================================
 * Part 6 -- several GPUs behind the boundary.  The reference fans its games out over every core INSIDE the
 * library (AsyncAgent::run, agent/async_agent/mod.rs:100-137, entered from cppbind/libcppconnect6/src/lib.rs:110-122) and gathers the
 * results over an mpsc channel (:129-134).  A c6_group is that for GPUs: one engine and one host thread per device, games sharded by uid
 * (member r of n plays uids [first_r, first_r + games_r) of the SAME seeded game set a single engine would play -- a recycled slot advances
 * by the total -- so the union over the members does not depend on how many devices there are), no exchange while playing, and the
 * training tuples pooled over the members by an RCCL all-gather (ncclAllGather over xGMI, one communicator per member from
 * ncclCommInitAll; librccl is opened, and the communicators are made, when a group first enables tuples).  Members that share a
 * device (devices = {0, 0}: a one-GPU rehearsal) pool through host memory instead -- RCCL admits one rank per device.
 * cpp_self_play itself uses a group when more than one device is visible and the call brings enough games (environment: C6_DEVICES =
 * "all" | "0,1,.." | unset; C6_GROUP_MIN_GAMES = games per device below which fewer devices are used, default 64).
 * ===================================================================================================================== */
typedef struct c6_group c6_group;

/* cfg->num_games = games over ALL members; cfg->device / game_uid_base / game_uid_stride are ignored (the group sets them per member).
 * devices[ndev]: HIP ordinals, repeats allowed (rehearsal).  ndev <= 0 or devices NULL: every visible device. */
int c6_group_create(const c6_config *cfg, const int *devices, int ndev, c6_group **out);
void c6_group_destroy(c6_group *g);
int c6_group_size(const c6_group *g);
/* member r's engine / device / shard (first uid, games): the engine is a full c6_engine (c6_engine_root, _stats, ... work on it) */
c6_engine *c6_group_engine(c6_group *g, int member);
int c6_group_member_info(const c6_group *g, int member, int *device, uint32_t *first_uid, int *games);
/* after c6_group_enable_tuples (the communicators are created there) -- 1: the members' tuple pooling runs on RCCL (all devices distinct and
 * librccl loaded); 0: through host memory */
int c6_group_uses_rccl(const c6_group *g);

int c6_group_load_weights(c6_group *g, const float *blob, size_t nfloats);   /* every member (weights are replicated: 14-34 MB) */
/* C6_EVAL_CALLBACK: the members' host threads call `cb` -- one at a time unless concurrent != 0 (the reference calls it from all
 * pool threads with no lock, lib.rs:110-122; serialised is the stricter, always-compatible default) */
int c6_group_set_callback(c6_group *g, c6_callback cb, int concurrent);

/* c6_engine_step / c6_engine_run on every member at once, one host thread per member; returns the first non-OK status (message:
 * c6_last_error() of the calling thread) after ALL members have returned */
int c6_group_step(c6_group *g, int rounds);
int c6_group_run(c6_group *g, long max_rounds);
int c6_group_sync(c6_group *g);
/* sums over the members (max_* fields: maxima) */
int c6_group_stats(c6_group *g, c6_stats *out);
/* finished games of all members, member by member (AsyncAgent::run hands results back in completion order, i.e. any order) */
int c6_group_fetch_games(c6_group *g, int cap, int *count, int *winners, int *lens, uint32_t *uids, int16_t *moves);

/* training tuples: every member records into group-owned double buffers of `capacity_records` each */
int c6_group_enable_tuples(c6_group *g, size_t capacity_records);
/* Pools what the members recorded since the last pool: afterwards EVERY member's device holds the same [ndev][padded][stride] array
 * (c6_group_pooled), counts[r] of member r's segment valid.  _begin retires the buffers (the engines continue into the other set at
 * once: rounds enqueued after it overlap the exchange), _end waits for the retired rounds, reads the counts and runs the all-gather on
 * the members' side streams.  c6_group_pool_tuples = _begin + _end.  dropped: records beyond capacity_records since the last pool. */
int c6_group_pool_begin(c6_group *g);
int c6_group_pool_end(c6_group *g, uint32_t *counts, size_t *padded, uint64_t *dropped);
int c6_group_pool_tuples(c6_group *g, uint32_t *counts, size_t *padded, uint64_t *dropped);
/* member r's copy of the last pool (device memory on its device; valid until the next pool) */
const void *c6_group_pooled(const c6_group *g, int member);
/* the valid records of member r's copy, compacted, to the host; *count = records copied (all members' segments, in member order) */
int c6_group_fetch_pooled(c6_group *g, int member, size_t cap, size_t *count, void *host_buf);
/* ... and device to device into a replay buffer living on member r's device (ReplayBuffer::push_game for every record) */
int c6_group_push_pooled(c6_group *g, int member, c6_replay *r);

#ifdef __cplusplus
}
#endif
#endif

/*
 * c6_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the self-play hot path of revsic/AlphaZero-Connect6
 * (Rust, `Connect6/libconnect6`).  Every function cites the reference
 * file:line it follows (paths relative to the reference checkout).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product (alphazero-connect6_amd/csrc) never does.
 *
 * Parity status
 *   - board / winner / game / simulate / augment / agent logic: PINNED against
 *     the reference's own known-answer tests (tests/test_oracle_*.py transcribe
 *     search_winner/tests.rs, game_impl/tests.rs, agent_impl/tests.rs,
 *     simulate/tests.rs, augment_tests.rs, alphazero_policy/tests.rs).
 *   - sampled quantities (Dirichlet noise, tie-breaks, RandomEvaluator): the
 *     reference uses unseeded `thread_rng` (rand 0.6, no lock file) and pins no
 *     numeric result => "parity unpinned" for the random STREAM.  The oracle
 *     uses the documented counter-based stream "C6RNG" (Philox4x32-10, see
 *     DESIGN.md) with the same distributions as the reference.
 *
 * The board side N is a run-time parameter (9, 15 or 19); the reference
 * compiles N = 15 (libconnect6/src/lib.rs:21).
 */
#ifndef C6_ORACLE_H
#define C6_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C6O_MAXN 19
#define C6O_MAXCELLS (C6O_MAXN * C6O_MAXN)

/* game/player/mod.rs:8-12 */
#define C6O_BLACK (-1)
#define C6O_NONE 0
#define C6O_WHITE 1

/* ---- game ------------------------------------------------------------ */

/* game/game_impl/mod.rs:117-121 */
typedef struct c6o_game {
    int n;
    int turn;
    int num_remain;
    int8_t board[C6O_MAXCELLS]; /* row-major n*n */
} c6o_game;

/* game/game_impl/mod.rs:26-62 */
typedef struct c6o_set_result {
    int player;
    int num_remain;
    int row, col;
} c6o_set_result;

enum { C6O_OK = 0, C6O_ERR_INVALID_POSITION = 1, C6O_ERR_ALREADY_SET = 2 };

void c6o_game_new(c6o_game *g, int n);                                   /* game_impl/mod.rs:125-131 */
int c6o_game_set(c6o_game *g, int row, int col, c6o_set_result *out);    /* game_impl/mod.rs:148-169 */
int c6o_game_is_end(const c6o_game *g);                                  /* game_impl/mod.rs:249-252 */
const char *c6o_strerror(int err);                                       /* game_impl/mod.rs:66-100 */
int c6o_search(const int8_t *board, int n);                              /* search_winner/mod.rs:201-246 */
int c6o_player_switch(int p);                                            /* player/mod.rs:24-30 */

/* ---- simulate -------------------------------------------------------- */

/* policy/simulate/mod.rs:33-36,93-98 (board + possible list + turn state) */
typedef struct c6o_sim {
    int n;
    int turn;
    int num_remain;
    int npossible;
    int16_t possible[C6O_MAXCELLS]; /* cell = row*n+col, in the reference's order */
    int8_t board[C6O_MAXCELLS];
} c6o_sim;

void c6o_sim_from_game(c6o_sim *s, const c6o_game *g); /* simulate/mod.rs:120-127 */
int c6o_sim_validate(const c6o_sim *s, int row, int col); /* :173-183 */
int c6o_sim_next_turn(const c6o_sim *s);                  /* :197-203 */
void c6o_sim_simulate_in(c6o_sim *s, int row, int col);   /* :262-274 */
void c6o_sim_rollback_in(c6o_sim *s, int row, int col);   /* :290-301 */
int c6o_diff_board(const int8_t *b1, const int8_t *b2, int n); /* default_policy/mod.rs:74-83; -1 if equal */

/* ---- augment --------------------------------------------------------- */

void c6o_rotate_left_i8(int8_t *b, int n);    /* augment.rs:7-15 */
void c6o_rotate_right_i8(int8_t *b, int n);   /* augment.rs:17-25 */
void c6o_flip_vertical_i8(int8_t *b, int n);  /* augment.rs:27-35 */
void c6o_flip_horizontal_i8(int8_t *b, int n);/* augment.rs:37-47 */
void c6o_rotate_left_f32(float *b, int n);
void c6o_rotate_right_f32(float *b, int n);
void c6o_flip_vertical_f32(float *b, int n);
void c6o_augment_way8(const int8_t *board, int n, int8_t *out8 /* 8*n*n */);   /* augment.rs:60-73 */
void c6o_recover_way8(float *probs8 /* 8*n*n, clobbered */, int n, float *out); /* augment.rs:75-92 */

/* ---- C6RNG: counter based stream (Philox4x32-10) ---------------------- */

enum {
    C6O_TAG_NOISE = 1, /* Dirichlet noise, ctr = (cell, attempt*2+{0,1}, search, tag | depth<<8) */
    C6O_TAG_TIE = 2,   /* tie-break,       ctr = (0, 0, search, tag | depth<<8) */
    C6O_TAG_EVAL_V = 3,/* RandomEvaluator value,   ctr = (image, 0, search, tag) word 0 */
    C6O_TAG_EVAL_P = 4,/* RandomEvaluator policy,  ctr = (cell>>2, image, search, tag) word cell&3 */
    C6O_TAG_SAMPLE = 5,/* replay sampling,         ctr = (i, draw_id, 0, tag) word 0 */
    C6O_TAG_ROLL = 6,  /* DefaultPolicy expand + playout picks, ctr = (draw>>2, 0, search, tag) word draw&3 */
    C6O_TAG_ROLL_POLICY = 7 /* DefaultPolicy::policy fallback,  ctr = (0, 0, searches so far, tag) word 0 */
};
void c6o_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void c6o_philox4x32_7(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);      /* the noise draws (C6RNG v5) */
float c6o_log4(float x);            /* log / exp of the Gamma sampler (C6RNG v4): fused polynomial forms, no division */
float c6o_exp4(float t, int *k);    /* exp(t) = return value * 2^k */
void c6o_gamma_draw4(double alpha, uint32_t k0, uint32_t k1, uint32_t cell, uint32_t search, uint32_t depth, float *m, int *k); /* variate m 2^k */
/* Gamma(alpha<1,1) draw number `cell` of select step (search, depth) of a game */
double c6o_gamma_draw(double alpha, uint32_t k0, uint32_t k1, uint32_t cell, uint32_t search, uint32_t depth);
/* sum of 64 lane-partials with the xor-butterfly order of the spec */
float c6o_butterfly_sum64f(const float part[64]);
/* Dirichlet noise for the empty cells of `board` (noise[cell], 0 elsewhere) */
void c6o_dirichlet_noise(double alpha, uint32_t k0, uint32_t k1, uint32_t search, uint32_t depth,
                         const int8_t *board, int n, float *noise);

/* ---- evaluator ------------------------------------------------------- */

/* policy/alphazero_policy/mod.rs:128-134.  boards: len*n*n int8 (-1/0/1);
 * values[len]; policies[len*n*n].  Return 0 on success. */
typedef int (*c6o_eval_fn)(void *ctx, int turn, int n, int len, const int8_t *boards, float *values,
                           float *policies);

/* the reference's C callback shape, cppbind/.../cppeval/mod.rs:8-13 */
typedef void (*c6o_c_callback)(int player, float *values, float *policies, int len);

/* ---- AlphaZero policy ------------------------------------------------ */

/* policy/alphazero_policy/mod.rs:91-96 */
typedef struct c6o_hyper {
    int num_simulation;
    float epsilon;
    double dirichlet_alpha;
    float c_puct;
} c6o_hyper;

typedef struct c6o_az c6o_az;

enum { C6O_EVAL_CUSTOM = 0, C6O_EVAL_RANDOM = 1, C6O_EVAL_UNIFORM = 2 };

/* eval_kind RANDOM: RandomEvaluator (mod.rs:137-169) on the C6RNG stream;
 * UNIFORM: value 0, policy 1/n^2; CUSTOM: fn/ctx. */
c6o_az *c6o_az_new(int n, const c6o_hyper *hp, int eval_kind, c6o_eval_fn fn, void *ctx, uint64_t seed,
                   uint32_t game_uid);
void c6o_az_free(c6o_az *az);
int c6o_az_next(c6o_az *az, const c6o_game *g);   /* mod.rs:464-489; returns cell or -1 */
void c6o_az_init(c6o_az *az, const c6o_sim *s);   /* mod.rs:302-306 */
int c6o_az_select(c6o_az *az, const c6o_sim *s);  /* mod.rs:313-323; cell or -1 */
void c6o_az_expand(c6o_az *az, const c6o_sim *s); /* mod.rs:331-381 */
void c6o_az_update(c6o_az *az, const c6o_sim *s, const int16_t *path, int len); /* mod.rs:386-408 */
void c6o_az_search(c6o_az *az, const c6o_sim *s); /* mod.rs:411-426 */
int c6o_az_policy(c6o_az *az, const c6o_sim *s);  /* mod.rs:433-452; cell or -1 */
void c6o_az_prune(c6o_az *az, const c6o_sim *s);  /* mod.rs:473-487 */
size_t c6o_az_map_len(const c6o_az *az);
uint32_t c6o_az_search_count(const c6o_az *az);
double c6o_az_mean_depth(const c6o_az *az);

/* node inspection (mod.rs:26-36) */
typedef struct c6o_node_info {
    int found;
    int visit;
    float value, q_sum, q_value, n_prob;
    int num_player;
    int n_next;
    int has_prob;
} c6o_node_info;
void c6o_az_node_info(const c6o_az *az, const int8_t *board, c6o_node_info *out);
/* copies node.prob (zeros if never expanded); returns 0 if node missing */
int c6o_az_node_prob(const c6o_az *az, const int8_t *board, float *prob);
/* per-cell child stats of the node for `board`: visit (-1 where no child), q_value, n_prob */
int c6o_az_children(const c6o_az *az, const int8_t *board, int *visit, float *q_value, float *n_prob);

/* ---- agent ----------------------------------------------------------- */

/* agent/agent_impl/mod.rs:34-44 */
typedef struct c6o_path {
    int turn;
    int row, col;
    int8_t board[C6O_MAXCELLS];
} c6o_path;

typedef struct c6o_play_result {
    int winner;
    int len;
    int err; /* C6O_OK or error from Game::set (agent_impl/mod.rs:163) */
    c6o_path *path; /* malloc'ed, len entries */
    int32_t *pi;    /* optional: len * n*n root child visit counts (build-side extension), may be NULL */
} c6o_play_result;

/* generic policy (policy/mod.rs:45-48): returns 1 and (*row,*col), or 0 for "no move" */
typedef int (*c6o_policy_fn)(void *ctx, const c6o_game *g, int *row, int *col);
void c6o_agent_play(int n, c6o_policy_fn fn, void *ctx, c6o_play_result *out); /* agent_impl/mod.rs:121-178 */
void c6o_play_result_free(c6o_play_result *r);

/* AlphaZero self-play of `num` independent games on `threads` worker threads
 * (agent/async_agent/mod.rs:100-137). game i uses game_uid = uid_base + i.
 * want_pi: also record root child visit counts per move. */
void c6o_self_play(int n, const c6o_hyper *hp, int eval_kind, c6o_eval_fn fn, void *ctx, uint64_t seed,
                   uint32_t uid_base, int num, int threads, int want_pi, c6o_play_result *out /* num */);

/* ---- replay buffer (AlphaZero/cpp_weighted/ReplayBuffer.hpp:10-73) ---- */
typedef struct c6o_replay c6o_replay;
c6o_replay *c6o_replay_new(int n, size_t max_size);                                         /* :12-20 */
void c6o_replay_free(c6o_replay *r);
void c6o_replay_push_game(c6o_replay *r, int winner, const c6o_path *path, int len);         /* :22-29 */
void c6o_replay_clear(c6o_replay *r);                                                        /* :31-33 */
size_t c6o_replay_size(const c6o_replay *r);                                                 /* :35-37 */
size_t c6o_replay_tail(const c6o_replay *r);
/* :39-65; the reference draws gen() % num_data from a random_device-seeded engine (unpinned): here the index of
 * draw i is Philox(seed; i, draw_id, 0, tag 5).x % num_data (C6RNG) */
void c6o_replay_sample(const c6o_replay *r, int num_sample, uint64_t seed, uint32_t draw_id, float *winners, float *players,
                       float *boards, float *poses);

/* ---- policy/value net (AlphaZero/cpp_weighted/model.hpp:9-95) -------- */

typedef struct c6o_net c6o_net;
/* blob layout: see DESIGN.md "weight blob" (same blob the product loads) */
size_t c6o_net_blob_floats(int n, int channels, int num_block);
c6o_net *c6o_net_new(int n, int channels, int num_block, const float *blob);
void c6o_net_free(c6o_net *net);
/* forward on `len` boards (float -1/0/1, row-major) with scalar player;
 * policy[len*n*n] softmax probs, value[len] tanh */
void c6o_net_forward(const c6o_net *net, float player, const float *boards, int len, float *policy,
                     float *value);
/* c6o_eval_fn adaptor: ctx = c6o_net* */
int c6o_net_eval(void *ctx, int turn, int n, int len, const int8_t *boards, float *values, float *policies);

/* ---- DefaultPolicy: pure MCTS with random playouts (c6_default.c) ----- */

typedef struct c6o_default c6o_default;
c6o_default *c6o_default_new(int n, int num_iter, uint64_t seed, uint32_t uid); /* default_policy/mod.rs:95-108 */
void c6o_default_free(c6o_default *dp);
void c6o_default_init(c6o_default *dp, const c6o_sim *s);                       /* :110-113 */
int c6o_default_select(const c6o_default *dp, const c6o_sim *s);                /* :119-148, cell or -1 */
int c6o_default_expand(c6o_default *dp, const c6o_sim *s);                      /* :150-170, cell */
void c6o_default_update(c6o_default *dp, const c6o_sim *s, const int16_t *path, int len); /* :172-211 */
void c6o_default_search(c6o_default *dp, const c6o_game *g);                    /* :213-236 */
int c6o_default_policy(c6o_default *dp, const c6o_sim *s);                      /* :238-258, cell or -1 */
int c6o_default_next(c6o_default *dp, const c6o_game *g);                       /* :262-273, cell or -1 */
size_t c6o_default_map_len(const c6o_default *dp);
uint32_t c6o_default_search_count(const c6o_default *dp);
int c6o_default_node(const c6o_default *dp, const int8_t *board, int *visit, int *black_win, int *n_next, int cap,
                     int32_t *child_cell, int32_t *child_visit, int32_t *child_black_win);

#ifdef __cplusplus
}
#endif
#endif

/*
 * c6_default.c -- CPU ORACLE (test infrastructure, NOT product code) for the pure-MCTS
 * rollout policy, SURVEY 8f rank 4: policy/default_policy/mod.rs:84-275.
 *
 * Restated with every quirk of the reference kept:
 *   - the tree is a map  board -> Node{visit, black_win, next_node}  that lives as long as the policy (never pruned);
 *   - select (:119-148) walks to the child with the largest  +-black_win / (1 + visit)  (f32, "-" for White), the LAST
 *     of equal maxima (Iterator::max_by), and stops where that value is 0 (so White stops at any child Black never won);
 *   - expand (:150-170) INSERTS the child, which resets a node that already exists for that board (a repeated pick or a
 *     transposition), and pushes it to the parent's list again (duplicates stay);
 *   - update (:172-211) plays uniformly random moves to the end, drops the playout when the board fills without a
 *     winner, and adds (1, winner == Black) to every node on the path, the expanded one included;
 *   - policy (:248-258) is one more select at the root, or a uniformly random empty cell when that stops at once.
 * The reference draws from `thread_rng` (unseeded, nothing pinned) => "parity unpinned" for the random STREAM; here every
 * `choose` is  possible[(w * len) >> 32]  with w the next 32-bit word of C6RNG (Philox4x32-10, key = (seed_lo, seed_hi + uid)):
 *   expand + playout draws of one search: word d & 3 of block ctr = (d >> 2, 0, search, C6O_TAG_ROLL), d = 0, 1, 2, ...
 *   the policy's fallback:                word 0 of ctr = (0, 0, searches so far, C6O_TAG_ROLL_POLICY).
 * `possible` stays in row-major order at every `choose` (Simulate removes in place, simulate/mod.rs:262-274), so the
 * pick is the k-th empty cell in row-major order.
 * Deviation (documented): with no empty cell left the reference's `choose().unwrap()` panics; here search returns and
 * policy yields -1.
 */
#include "c6_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct dnode {
    uint64_t h;
    int visit, black_win;
    int n_next, cap_next;
    struct dnode **next; /* push order; the reference stores the hashes, the node of a hash never moves */
    int8_t board[C6O_MAXCELLS];
} dnode;

struct c6o_default {
    int n, cells, num_iter;
    uint32_t k0, k1;
    uint32_t nsearch; /* searches started so far */
    uint32_t cur;     /* C6RNG `search` coordinate of the running search */
    uint32_t ndraw;   /* draws taken inside it */
    uint64_t zob[2][C6O_MAXCELLS];
    dnode **tab;
    size_t cap, len;
};

static uint64_t mix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* stands in for default_policy/mod.rs:55-59 (SipHash of the board); boards are compared on every hit */
static uint64_t dp_hash(const c6o_default *dp, const int8_t *board) {
    uint64_t h = 0x7654321ULL;
    for (int i = 0; i < dp->cells; i++)
        if (board[i] != C6O_NONE) h ^= dp->zob[board[i] == C6O_WHITE][i];
    return h;
}

static dnode *dp_get(const c6o_default *dp, const int8_t *board) {
    uint64_t h = dp_hash(dp, board);
    size_t mask = dp->cap - 1, i = (size_t)h & mask;
    while (dp->tab[i]) {
        dnode *nd = dp->tab[i];
        if (nd->h == h && memcmp(nd->board, board, (size_t)dp->cells) == 0) return nd;
        i = (i + 1) & mask;
    }
    return NULL;
}

static void dp_put_raw(dnode **tab, size_t cap, dnode *nd) {
    size_t mask = cap - 1, i = (size_t)nd->h & mask;
    while (tab[i]) i = (i + 1) & mask;
    tab[i] = nd;
}

/* HashMap::insert: a fresh Node::new(board) (:28-37) under the board's key, replacing what was there */
static dnode *dp_insert_fresh(c6o_default *dp, const int8_t *board) {
    dnode *nd = dp_get(dp, board);
    if (nd) {
        nd->visit = nd->black_win = 0;
        nd->n_next = 0;
        return nd;
    }
    if ((dp->len + 1) * 2 > dp->cap) {
        size_t ncap = dp->cap * 2;
        dnode **nt = (dnode **)calloc(ncap, sizeof(dnode *));
        for (size_t i = 0; i < dp->cap; i++)
            if (dp->tab[i]) dp_put_raw(nt, ncap, dp->tab[i]);
        free(dp->tab);
        dp->tab = nt;
        dp->cap = ncap;
    }
    nd = (dnode *)calloc(1, sizeof(dnode));
    nd->h = dp_hash(dp, board);
    memcpy(nd->board, board, (size_t)dp->cells);
    dp_put_raw(dp->tab, dp->cap, nd);
    dp->len++;
    return nd;
}

/* DefaultPolicy::with_num_iter :103-108 (new() :95-100 is num_iter = 50) */
c6o_default *c6o_default_new(int n, int num_iter, uint64_t seed, uint32_t uid) {
    c6o_default *dp = (c6o_default *)calloc(1, sizeof *dp);
    dp->n = n;
    dp->cells = n * n;
    dp->num_iter = num_iter;
    dp->k0 = (uint32_t)seed;
    dp->k1 = (uint32_t)(seed >> 32) + uid;
    uint64_t s = 0xD1CE5EEDULL;
    for (int p = 0; p < 2; p++)
        for (int i = 0; i < C6O_MAXCELLS; i++) dp->zob[p][i] = mix64(&s);
    dp->cap = 1024;
    dp->tab = (dnode **)calloc(dp->cap, sizeof(dnode *));
    return dp;
}

void c6o_default_free(c6o_default *dp) {
    if (!dp) return;
    for (size_t i = 0; i < dp->cap; i++)
        if (dp->tab[i]) {
            free(dp->tab[i]->next);
            free(dp->tab[i]);
        }
    free(dp->tab);
    free(dp);
}

/* `choose`: index into a list of `len` entries from the next word of the running search's stream */
static int dp_draw(c6o_default *dp, int len) {
    uint32_t d = dp->ndraw++;
    uint32_t ctr[4] = {d >> 2, 0, dp->cur, C6O_TAG_ROLL}, key[2] = {dp->k0, dp->k1}, w[4];
    c6o_philox4x32_10(ctr, key, w);
    return (int)(((uint64_t)w[d & 3] * (uint64_t)len) >> 32);
}

/* :110-113 */
void c6o_default_init(c6o_default *dp, const c6o_sim *s) {
    if (!dp_get(dp, s->board)) dp_insert_fresh(dp, s->board);
}

/* :119-148; returns the cell or -1 */
int c6o_default_select(const c6o_default *dp, const c6o_sim *s) {
    const dnode *tn = dp_get(dp, s->board);
    if (!tn) abort(); /* unwrap() */
    const dnode *best = NULL;
    float bestp = 0.f;
    for (int i = 0; i < tn->n_next; i++) {
        const dnode *c = tn->next[i];
        float p = (float)c->black_win / (1.f + (float)c->visit);
        if (s->turn == C6O_WHITE) p = -p;
        if (!best || !(bestp > p)) { /* max_by keeps the later of equal elements */
            best = c;
            bestp = p;
        }
    }
    if (best && bestp != 0.f) return c6o_diff_board(best->board, s->board, dp->n);
    return -1;
}

/* :150-170; returns the cell */
int c6o_default_expand(c6o_default *dp, const c6o_sim *s) {
    int cell = s->possible[dp_draw(dp, s->npossible)];
    int8_t board[C6O_MAXCELLS];
    memcpy(board, s->board, (size_t)dp->cells);
    board[cell] = (int8_t)s->turn;
    dnode *child = dp_insert_fresh(dp, board);
    dnode *parent = dp_get(dp, s->board);
    if (!parent) abort(); /* unwrap() */
    if (parent->n_next == parent->cap_next) {
        parent->cap_next = parent->cap_next ? parent->cap_next * 2 : 4;
        parent->next = (dnode **)realloc(parent->next, sizeof(dnode *) * (size_t)parent->cap_next);
    }
    parent->next[parent->n_next++] = child;
    return cell;
}

/* :172-211; path = cells from the root to s (the expanded move last) */
void c6o_default_update(c6o_default *dp, const c6o_sim *s, const int16_t *path, int len) {
    c6o_sim roll = *s;
    while (c6o_search(roll.board, roll.n) == C6O_NONE) {
        if (roll.npossible == 0) break;
        int cell = roll.possible[dp_draw(dp, roll.npossible)];
        c6o_sim_simulate_in(&roll, cell / roll.n, cell % roll.n);
    }
    int win = c6o_search(roll.board, roll.n);
    if (win == C6O_NONE) return;
    int black_win = win == C6O_BLACK;

    c6o_sim back = *s;
    dnode *nd = dp_get(dp, back.board);
    if (!nd) abort();
    nd->visit += 1;
    nd->black_win += black_win;
    for (int i = len - 1; i >= 0; i--) {
        c6o_sim_rollback_in(&back, path[i] / back.n, path[i] % back.n);
        nd = dp_get(dp, back.board);
        if (!nd) abort();
        nd->visit += 1;
        nd->black_win += black_win;
    }
}

/* :213-236 */
void c6o_default_search(c6o_default *dp, const c6o_game *g) {
    dp->cur = dp->nsearch++;
    dp->ndraw = 0;
    c6o_sim s;
    c6o_sim_from_game(&s, g);
    c6o_default_init(dp, &s);
    int16_t path[C6O_MAXCELLS];
    int len = 0, cell;
    while ((cell = c6o_default_select(dp, &s)) >= 0) {
        path[len++] = (int16_t)cell;
        c6o_sim_simulate_in(&s, cell / s.n, cell % s.n);
    }
    if (c6o_search(s.board, s.n) != C6O_NONE) return;
    if (s.npossible == 0) return; /* the reference panics here (choose on an empty list) */
    cell = c6o_default_expand(dp, &s);
    path[len++] = (int16_t)cell;
    c6o_sim_simulate_in(&s, cell / s.n, cell % s.n);
    c6o_default_update(dp, &s, path, len);
}

/* :238-258; returns the cell or -1 */
int c6o_default_policy(c6o_default *dp, const c6o_sim *s) {
    int cell = c6o_default_select(dp, s);
    if (cell >= 0) return cell;
    if (s->npossible == 0) return -1; /* the reference panics here */
    uint32_t ctr[4] = {0, 0, dp->nsearch, C6O_TAG_ROLL_POLICY}, key[2] = {dp->k0, dp->k1}, w[4];
    c6o_philox4x32_10(ctr, key, w);
    return s->possible[(int)(((uint64_t)w[0] * (uint64_t)s->npossible) >> 32)];
}

/* Policy::next :262-273 */
int c6o_default_next(c6o_default *dp, const c6o_game *g) {
    for (int i = 0; i < dp->num_iter; i++) c6o_default_search(dp, g);
    c6o_sim s;
    c6o_sim_from_game(&s, g);
    if (!dp_get(dp, s.board)) return -1; /* num_iter == 0 and never initialised: the reference's unwrap() panics */
    return c6o_default_policy(dp, &s);
}

size_t c6o_default_map_len(const c6o_default *dp) { return dp->len; }
uint32_t c6o_default_search_count(const c6o_default *dp) { return dp->nsearch; }

/* test access: the node of `board` (0 if absent) and its children in push order */
int c6o_default_node(const c6o_default *dp, const int8_t *board, int *visit, int *black_win, int *n_next, int cap,
                     int32_t *child_cell, int32_t *child_visit, int32_t *child_black_win) {
    const dnode *nd = dp_get(dp, board);
    if (!nd) return 0;
    *visit = nd->visit;
    *black_win = nd->black_win;
    *n_next = nd->n_next;
    for (int i = 0; i < nd->n_next && i < cap; i++) {
        child_cell[i] = c6o_diff_board(nd->next[i]->board, board, dp->n);
        child_visit[i] = nd->next[i]->visit;
        child_black_win[i] = nd->next[i]->black_win;
    }
    return 1;
}

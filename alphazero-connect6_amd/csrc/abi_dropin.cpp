// abi_dropin.cpp -- the reference's C ABI (Connect6/cppbind/libcppconnect6, N = 15) on top of the
// GPU engine.  Symbol for symbol what Connect6/cppbind/headers/connect6.hpp:44-93 declares, so the
// reference's C++ wrapper, samples and trainers link against libconnect6_amd.so unchanged.
#include "../../include/connect6_amd.h"
#include "c6_engine.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static const int N = C6_REF_BOARD_SIZE, CELLS = C6_REF_BOARD_CAPACITY;

static void die(const char *what) {
    // the reference panics across the FFI on errors (lib.rs:40,107 `unwrap()`); never unwind through C
    std::fprintf(stderr, "libconnect6_amd: %s: %s\n", what, c6_last_error());
    std::abort();
}

// the engine's entry points select their device on the calling thread: a caller that uses HIP itself (a libtorch trainer) finds its own current device again
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

static uint64_t pick_seed() {
    if (const char *s = std::getenv("C6_SEED")) return std::strtoull(s, nullptr, 0);
    std::random_device rd;   // the reference seeds from the OS (thread_rng)
    return ((uint64_t)rd() << 32) | rd();
}

// RawPath::with_path + RawPlayResult::with_result (rawobj/mod.rs:66-107): replay the stones on the
// host to rebuild "board BEFORE the move" (agent/agent_impl/mod.rs:139-143)
static c6_play_result make_result(int winner, int len, const int16_t *moves, c6_alloc_path alloc_path) {
    c6_play_result r;
    r.winner = winner;
    r.len = len;
    r.paths = alloc_path(len);
    int board[C6_REF_BOARD_CAPACITY] = {0};
    int turn = -1, remain = 1;   // Game::new, game_impl/mod.rs:125-131
    for (int k = 0; k < len; k++) {
        c6_path &p = r.paths[k];
        p.turn = turn;
        std::memcpy(p.board, board, sizeof board);
        p.row = moves[k] / N;
        p.col = moves[k] % N;
        board[moves[k]] = turn;
        if (--remain <= 0) { remain = 2; turn = -turn; }   // game_impl/mod.rs:160-167
    }
    return r;
}

static void print_board(const int8_t *board);
std::vector<int> c6_dropin_devices(int num_games);      // group.cpp: C6_DEVICES / C6_GROUP_MIN_GAMES

// "{player:?} ({row}, {col}), remain {n}, {s}.{ms} elapsed", agent_impl/mod.rs:147-160 (num_remain: after the decrement,
// before the reset -- SetResult, game_impl/mod.rs:158-160)
static void print_stone(int turn, int pos, int remain_before, long long ms) {
    std::printf("%s (%c, %c), remain %d, %lld.%lld elapsed\n", turn == -1 ? "Black" : "White", 'a' + pos / N, 'A' + pos % N, remain_before - 1,
                ms / 1000, ms % 1000);
    std::fflush(stdout);
}

extern "C" c6_vec cpp_self_play(c6_callback callback, c6_alloc_path alloc_path, c6_alloc_result alloc_result, int num_simulation,
                                float epsilon, double dirichlet_alpha, float c_puct, bool debug, int num_game_thread) {
    DeviceGuard dg_;
    c6_config cfg;
    c6_default_config(&cfg);
    cfg.board_size = N;
    cfg.num_games = num_game_thread < 1 ? 1 : num_game_thread;
    cfg.num_simulation = num_simulation;
    cfg.epsilon = epsilon;
    cfg.dirichlet_alpha = dirichlet_alpha;
    cfg.c_puct = c_puct;
    cfg.seed = pick_seed();
    cfg.evaluator = C6_EVAL_CALLBACK;
    // the reference ABI has no field for the engine's pool sizes: environment knobs (defaults: 4 * num_simulation + 64 records
    // per game, doubled on demand up to 65000)
    if (const char *r = std::getenv("C6_RECORDS")) cfg.records_per_game = std::atoi(r);
    if (const char *r = std::getenv("C6_MAX_RECORDS")) cfg.max_records_per_game = std::atoi(r);
    const int G = cfg.num_games;
    const auto t_start = std::chrono::steady_clock::now();
    // AsyncAgent::run fans the games out over every core inside the library (async_agent/mod.rs:100-137, lib.rs:110-122): here over every visible GPU
    // -- a c6_group (group.cpp: one engine + host thread per device, games sharded by uid) -- when the call brings enough games to fill more than one
    const std::vector<int> devs = c6_dropin_devices(G);
    if (devs.size() > 1) {
        c6_group *grp = nullptr;
        if (c6_group_create(&cfg, devs.data(), (int)devs.size(), &grp) != C6_OK) die("cpp_self_play: group");
        const char *cc = std::getenv("C6_CALLBACK_CONCURRENT");
        c6_group_set_callback(grp, callback, cc && cc[0] == '1');
        const int grc = c6_group_run(grp, 0);
        if (grc == C6_ERR_OVERFLOW) std::fprintf(stderr, "libconnect6_amd: cpp_self_play: %s\n", c6_last_error());
        else if (grc != C6_OK) die("cpp_self_play: run");
        std::vector<int> winners(G), lens(G);
        std::vector<int16_t> moves((size_t)G * CELLS);
        int count = 0;
        if (c6_group_fetch_games(grp, G, &count, winners.data(), lens.data(), nullptr, moves.data()) != C6_OK) die("cpp_self_play: fetch");
        const long long ms_all = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
        c6_vec v;
        v.len = count;
        v.vec = alloc_result(count);
        for (int i = 0; i < count; i++) {
            v.vec[i] = make_result(winners[i], lens[i], &moves[(size_t)i * CELLS], alloc_path);
            if (debug) std::printf("run: %d, elapsed %lld.%llds\n", i, ms_all / 1000, ms_all % 1000);      // async_agent/mod.rs:117-124
        }
        c6_group_destroy(grp);
        return v;
    }
    cfg.device = devs[0];
    c6_engine *e = nullptr;
    if (c6_engine_create(&cfg, &e) != C6_OK) die("cpp_self_play: engine");
    c6_engine_set_callback(e, callback);
    int rc = C6_OK;
    if (debug && G == 1) {
        // lib.rs:98-108: one game on the caller's thread through Agent::debug -- the board before every stone, one line per
        // stone, the final board (agent_impl/mod.rs:126-176).  The engine is stepped one move (= num_simulation rounds) at a time.
        std::vector<int8_t> board(CELLS, 0), after(CELLS, 0);
        for (;;) {
            c6_root_info info;
            if (c6_engine_root(e, 0, &info, board.data(), nullptr, nullptr, nullptr) != C6_OK) die("cpp_self_play: root");
            if (info.status != 0) break;
            print_board(board.data());
            const auto t0 = std::chrono::steady_clock::now();
            c6_root_info ai = info;
            while (rc == C6_OK && ai.status == 0 && ai.stones == info.stones) {      // one move = num_simulation rounds (more calls only
                rc = c6_engine_run(e, ai.stones == info.stones && ai.nsearch == info.nsearch ? num_simulation : 1);   // after a pool growth)
                if (rc == C6_OK && c6_engine_root(e, 0, &ai, after.data(), nullptr, nullptr, nullptr) != C6_OK) die("cpp_self_play: root");
            }
            if (rc != C6_OK) break;
            int pos = -1;
            for (int c = 0; c < CELLS; c++) if (after[c] != board[c]) pos = c;
            if (pos < 0) break;                         // the policy had no move (agent_impl/mod.rs:135-137) or the game stopped
            const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            print_stone(info.turn, pos, info.remain, ms);
        }
        if (rc == C6_OK) rc = c6_engine_run(e, 0);      // closes the books (flushes a game that stopped in an error state)
        c6_engine_root(e, 0, nullptr, board.data(), nullptr, nullptr, nullptr);
        print_board(board.data());
    } else {
        rc = c6_engine_run(e, 0);
    }
    if (rc == C6_ERR_OVERFLOW) {
        // The reference has no error channel here (SURVEY 8b "Errors": its unwrap() would abort the process).  The games that
        // stopped are handed back as they stand -- winner 0, the stones played so far -- and the reason goes to stderr.
        std::fprintf(stderr, "libconnect6_amd: cpp_self_play: %s\n", c6_last_error());
    } else if (rc != C6_OK) {
        die("cpp_self_play: run");
    }
    std::vector<int> winners(G), lens(G);
    std::vector<int16_t> moves((size_t)G * CELLS);
    int count = 0;
    if (c6_engine_fetch_games(e, G, &count, winners.data(), lens.data(), nullptr, moves.data()) != C6_OK) die("cpp_self_play: fetch");
    const long long ms_all = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
    c6_vec v;
    v.len = count;
    v.vec = alloc_result(count);                              // RawVec::with_vec, rawobj/mod.rs:112-123
    for (int i = 0; i < count; i++) {
        v.vec[i] = make_result(winners[i], lens[i], &moves[(size_t)i * CELLS], alloc_path);
        // AsyncAgent::debug, async_agent/mod.rs:117-124 (the games of a call run concurrently in one batch: one clock)
        if (debug && G > 1) std::printf("run: %d, elapsed %lld.%llds\n", i, ms_all / 1000, ms_all % 1000);
    }
    c6_engine_destroy(e);
    return v;
}

extern "C" c6_vec cpp_play(c6_policy_callback callback, c6_alloc_path alloc_path, c6_alloc_result alloc_result, bool debug,
                           int num_game_thread) {
    const int G = num_game_thread < 1 ? 1 : num_game_thread;
    // Agent::play with a CppPolicy (cpp_policy/mod.rs:22-43) for G games; Game::set / is_game_end on the GPU
    std::vector<int8_t> boards((size_t)CELLS * G, 0), turn(G, -1), remain(G, 1), winner(G, 0), err(G, 0);
    std::vector<int32_t> pos(G);
    std::vector<std::vector<int16_t>> hist(G);
    std::vector<char> done(G, 0);
    std::vector<int> result(G, 0);
    int live = G;
    const bool trace = debug && G == 1;      // lib.rs:32-41: one game through Agent::debug
    const auto t_start = std::chrono::steady_clock::now();
    auto col0 = [&](int g) { std::vector<int8_t> b(CELLS); for (int c = 0; c < CELLS; c++) b[c] = boards[(size_t)c * G + g]; return b; };
    while (live > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        const int turn0 = turn[0], remain0 = remain[0];
        if (trace) print_board(col0(0).data());
        for (int g = 0; g < G; g++) {
            pos[g] = -1;
            if (done[g]) continue;
            float fb[C6_REF_BOARD_CAPACITY];
            for (int c = 0; c < CELLS; c++) fb[c] = (float)boards[(size_t)c * G + g];   // board_to_float, cpp_support.rs:10-18
            int p[2] = {-1, -1};
            callback(fb, p);
            if (p[0] == -1) {            // "no move" -> game ends as a draw, agent_impl/mod.rs:135-137
                done[g] = 1;
                live--;
                continue;
            }
            // rows/cols outside the board become an out-of-range cell: "invalid position"
            const bool inside = p[0] >= 0 && p[0] < N && p[1] >= 0 && p[1] < N;
            pos[g] = inside ? p[0] * N + p[1] : CELLS;
            hist[g].push_back((int16_t)(inside ? pos[g] : 0));
        }
        if (live == 0) break;
        if (c6_board_step(N, G, boards.data(), turn.data(), remain.data(), pos.data(), winner.data(), err.data()) != C6_OK) die("cpp_play");
        for (int g = 0; g < G; g++) {
            if (done[g] || pos[g] < 0) continue;
            if (err[g] != C6_OK) {       // the reference returns Err and `unwrap()` aborts (lib.rs:40)
                std::fprintf(stderr, "libconnect6_amd: cpp_play: %s\n", err[g] == C6_ERR_ALREADY_SET ? "already set position" : "invalid position");
                std::abort();
            }
            if (winner[g] != 0) { result[g] = winner[g]; done[g] = 1; live--; }
        }
        if (trace && pos[0] >= 0)
            print_stone(turn0, pos[0], remain0, std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count());
    }
    if (trace) print_board(col0(0).data());
    const long long ms_all = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
    c6_vec v;
    v.len = G;
    v.vec = alloc_result(G);
    for (int g = 0; g < G; g++) {
        v.vec[g] = make_result(result[g], (int)hist[g].size(), hist[g].data(), alloc_path);
        if (debug && G > 1) std::printf("run: %d, elapsed %lld.%llds\n", g, ms_all / 1000, ms_all % 1000);      // async_agent/mod.rs:117-124
    }
    return v;
}

// Game::print, game_impl/mod.rs:217-237
static void print_board(const int8_t *board) {
    std::printf("0 A B C D E F G H I J K L M N O\n");
    for (int i = 0; i < N; i++) {
        std::printf("%c ", 'a' + i);
        for (int j = 0; j < N; j++) std::printf("%s", board[i * N + j] == -1 ? "X " : (board[i * N + j] == 1 ? "O " : "_ "));
        std::printf("\n");
    }
    std::fflush(stdout);
}

// IoPolicy::next, policy/io_policy/mod.rs:70-107: up to 10 bytes per read, the alphabetic characters must be exactly a
// lower-case row and an upper-case column inside the board; otherwise "invalid input, retry".  EOF ends the game (the
// reference would spin on empty reads).
static bool read_human_move(int *row, int *col) {
    for (;;) {
        char buf[10];
        const size_t got = std::fread(buf, 1, 1, stdin);        // line-wise below: a terminal delivers whole lines
        if (got == 0) return false;
        size_t n = 1;
        while (n < sizeof buf && buf[n - 1] != '\n') {
            if (std::fread(buf + n, 1, 1, stdin) != 1) break;
            n++;
        }
        char q[10];
        int k = 0;
        for (size_t i = 0; i < n; i++)
            if ((buf[i] >= 'a' && buf[i] <= 'z') || (buf[i] >= 'A' && buf[i] <= 'Z')) q[k++] = buf[i];
        if (k == 2) {
            const unsigned r = (unsigned)(unsigned char)q[0] - 0x61u, c = (unsigned)(unsigned char)q[1] - 0x41u;   // usize arithmetic: wraps
            if (r < (unsigned)N && c < (unsigned)N) { *row = (int)r; *col = (int)c; return true; }
        }
        std::printf("invalid input, retry\n");
        std::fflush(stdout);
    }
}

// cpp_play_with, lib.rs:140-169: MultiPolicy(black = AlphaZero over the callback evaluator, white = IoPolicy on
// stdin/stdout), Agent::debug(...).play(): the board is printed before every stone and each stone is logged as
// "{player:?} ({row}, {col}), remain {n}, {s}.{ms} elapsed" (agent_impl/mod.rs:121-178).
extern "C" c6_play_result cpp_play_with(c6_callback callback, c6_alloc_path alloc_path, int num_simulation, float epsilon, double dirichlet_alpha,
                                        float c_puct) {
    DeviceGuard dg_;
    c6_config cfg;
    c6_default_config(&cfg);
    cfg.board_size = N;
    cfg.num_games = 1;
    cfg.num_simulation = num_simulation;
    cfg.epsilon = epsilon;
    cfg.dirichlet_alpha = dirichlet_alpha;
    cfg.c_puct = c_puct;
    cfg.seed = pick_seed();
    cfg.evaluator = C6_EVAL_CALLBACK;
    c6_engine *e = nullptr;
    if (c6_engine_create(&cfg, &e) != C6_OK) die("cpp_play_with: engine");
    c6_engine_set_callback(e, callback);
    std::vector<int8_t> board(CELLS, 0);
    std::vector<int16_t> hist;
    int winner = 0;
    for (;;) {
        c6_root_info info;
        if (c6_engine_root(e, 0, &info, board.data(), nullptr, nullptr, nullptr) != C6_OK) die("cpp_play_with: root");
        if (info.status != 0 /* not playing any more */) { winner = info.winner; break; }
        print_board(board.data());
        const auto t0 = std::chrono::steady_clock::now();
        int pos = -1;
        if (info.turn == -1) {                                   // Black: AlphaZero::next = num_simulation searches + policy()
            if (c6_engine_step(e, num_simulation) != C6_OK) die("cpp_play_with: search");
            std::vector<int8_t> after(CELLS);
            c6_root_info ai;
            if (c6_engine_root(e, 0, &ai, after.data(), nullptr, nullptr, nullptr) != C6_OK) die("cpp_play_with: root");
            for (int c = 0; c < CELLS; c++) if (after[c] != board[c]) pos = c;
            if (pos < 0) { winner = ai.winner; break; }         // the policy had no move: the game ends (agent_impl/mod.rs:135-137)
        } else {                                                 // White: the human
            int row, col;
            if (!read_human_move(&row, &col)) break;
            pos = row * N + col;
            const int32_t p = pos;
            if (c6_engine_apply_moves(e, &p) != C6_OK) die("cpp_play_with: move");
            c6_root_info ai;
            if (c6_engine_root(e, 0, &ai, nullptr, nullptr, nullptr, nullptr) != C6_OK) die("cpp_play_with: root");
            if (ai.status == 2 /* error: already set position */) {
                std::fprintf(stderr, "libconnect6_amd: cpp_play_with: already set position\n");   // Err -> unwrap() panic, lib.rs:167
                std::abort();
            }
        }
        hist.push_back((int16_t)pos);
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        const int remain_after = info.remain - 1;                // SetResult::num_remain: after the decrement, before the reset
        (void)remain_after;
        print_stone(info.turn, pos, info.remain, (long long)ms);
    }
    c6_engine_root(e, 0, nullptr, board.data(), nullptr, nullptr, nullptr);
    print_board(board.data());
    c6_engine_destroy(e);
    return make_result(winner, (int)hist.size(), hist.data(), alloc_path);
}

// ---- FFI echo hooks (ffi_test/mod.rs), host only ---------------------------------------------

extern "C" c6_path test_new_raw_path(void) {   // RawPath::new, rawobj/mod.rs:53-61
    c6_path p;
    std::memset(&p, 0, sizeof p);
    return p;
}

extern "C" c6_path test_with_raw_path(void) {  // ffi_test/mod.rs:19-34
    c6_path p;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) p.board[i][j] = (i * N + j) % 3 - 1;
    p.turn = 1;
    p.row = 0;
    p.col = N % 5 + 1;
    return p;
}

extern "C" c6_path test_echo_raw_path(int turn, int *board, int row, int col) {   // ffi_test/mod.rs:38-57
    c6_path p;
    p.turn = turn;
    std::memcpy(p.board, board, sizeof p.board);
    p.row = row;
    p.col = col;
    return p;
}

static int player_from(int v) { return v == -1 ? -1 : (v == 1 ? 1 : 0); }   // From<i32>, player/mod.rs:53-61

extern "C" c6_play_result test_with_raw_play_result(c6_alloc_path allocator) {   // ffi_test/mod.rs:61-86
    c6_play_result r;
    r.winner = -1;
    r.len = 10;
    r.paths = allocator(10);
    int player = -1;
    for (int i = 0; i < 10; i++) {
        c6_path &p = r.paths[i];
        std::memset(&p, 0, sizeof p);
        for (int j = 0; j < i + 1; j++) p.board[j][j] = player_from((i + j) % 3 - 1);
        p.turn = player;
        p.row = i;
        p.col = i + 1;
        player = -player;
    }
    return r;
}

extern "C" c6_play_result test_echo_raw_play_result(int winner, c6_path *path, int len, c6_alloc_path allocator) {   // :90-121
    c6_play_result r;
    r.winner = player_from(winner);
    r.len = len;
    r.paths = allocator(len);
    for (int i = 0; i < len; i++) {
        c6_path &p = r.paths[i];
        p.turn = player_from(path[i].turn);
        for (int j = 0; j < N; j++)
            for (int k = 0; k < N; k++) p.board[j][k] = player_from(path[i].board[j][k]);
        p.row = path[i].row;
        p.col = path[i].col;
    }
    return r;
}

extern "C" c6_vec_int test_with_raw_vec(c6_alloc_int allocator) {   // ffi_test/mod.rs:125-129
    c6_vec_int v;
    v.len = 6;
    v.vec = allocator(6);
    for (int i = 0; i < 6; i++) v.vec[i] = i;
    return v;
}

extern "C" c6_vec_int test_echo_raw_vec(int *ptr, int len, c6_alloc_int allocator) {   // ffi_test/mod.rs:133-145
    c6_vec_int v;
    v.len = len;
    v.vec = allocator(len);
    for (int i = 0; i < len; i++) v.vec[i] = ptr[i];
    return v;
}

extern "C" c6_vec_float test_echo_cppeval(int turn, int *boards, int len, c6_callback callback, c6_alloc_float allocator) {   // :149-197
    const int player = player_from(turn);
    std::vector<float> values((size_t)len, 0.f), policies((size_t)len * CELLS);
    for (size_t i = 0; i < policies.size(); i++) policies[i] = (float)player_from(boards[i]);   // CppEval::callback, cppeval/mod.rs:27-44
    callback(player, values.data(), policies.data(), len);
    c6_vec_float v;
    v.len = len + len * CELLS;
    v.vec = allocator(v.len);
    for (int i = 0; i < len; i++) v.vec[i] = values[i];
    for (size_t i = 0; i < policies.size(); i++) v.vec[len + i] = policies[i];
    return v;
}

extern "C" c6_vec_int test_cpp_policy(float *board, c6_policy_callback callback, c6_alloc_int allocator) {   // ffi_test/mod.rs:201-224
    float fb[C6_REF_BOARD_CAPACITY];
    for (int c = 0; c < CELLS; c++) fb[c] = (float)player_from((int)board[c]);
    int p[2] = {-1, -1};
    callback(fb, p);                           // CppPolicy::callback, cpp_policy/mod.rs:22-36
    c6_vec_int v;
    if (p[0] == -1) {
        v.len = 0;
        v.vec = allocator(0);
    } else {
        v.len = 2;
        v.vec = allocator(2);
        v.vec[0] = p[0];
        v.vec[1] = p[1];
    }
    return v;
}

// Self-play over several GPUs from a compiled C++ host, no Python and no launcher: the c6_group API of include/connect6_amd.h part 6 -- one engine and
// one host thread per device inside the library, games sharded by uid, training tuples pooled over the members by ncclAllGather (RCCL over xGMI).
// The reference's counterpart: AsyncAgent::run's thread pool + mpsc collection (agent/async_agent/mod.rs:100-137) behind cpp_self_play
// (cppbind/libcppconnect6/src/lib.rs:110-122), as AlphaZero/cpp_weighted/main.cpp:143 calls it.
//   g++ -O2 -std=c++17 -I include tools/cpp/group_selfplay.cpp -L alphazero-connect6_amd -lconnect6_amd -Wl,-rpath,'$ORIGIN/../../alphazero-connect6_amd' -o tools/cpp/_group_selfplay
//   tools/cpp/_group_selfplay DEVICES BOARD GAMES SIMS SEED [evaluator = random | uniform | net:BLOB.bin:CHANNELS:BLOCKS] [rounds = 0: to the end] [pool_every = 0]
//     DEVICES: "all" or a comma list of HIP ordinals (repeats allowed: "0,0" rehearses two members on one GPU, pooling through host memory)
//     rounds > 0: steady-state mode (recycle + stagger), `rounds` simulation rounds, tuples pooled every `pool_every` rounds with the exchange overlapped
// Prints one JSON line.
#include "connect6_amd.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void die(const char *what) {
    std::fprintf(stderr, "group_selfplay: %s: %s\n", what, c6_last_error());
    std::exit(1);
}

int main(int argc, char **argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s DEVICES BOARD GAMES SIMS SEED [evaluator] [rounds] [pool_every]\n", argv[0]); return 2; }
    std::vector<int> dev;
    if (std::strcmp(argv[1], "all") != 0)
        for (const char *p = argv[1]; *p;) { char *e; dev.push_back((int)std::strtol(p, &e, 10)); p = *e == ',' ? e + 1 : e; }
    const int n = std::atoi(argv[2]), games = std::atoi(argv[3]), sims = std::atoi(argv[4]);
    const unsigned long long seed = std::strtoull(argv[5], nullptr, 0);
    const std::string ev = argc > 6 ? argv[6] : "random";
    const long rounds = argc > 7 ? std::atol(argv[7]) : 0;
    const int pool_every = argc > 8 ? std::atoi(argv[8]) : 0;

    c6_config cfg;
    c6_default_config(&cfg);
    cfg.board_size = n; cfg.num_games = games; cfg.num_simulation = sims; cfg.seed = seed;
    cfg.evaluator = ev == "uniform" ? C6_EVAL_UNIFORM : ev.rfind("net:", 0) == 0 ? C6_EVAL_NET : C6_EVAL_RANDOM;
    cfg.recycle = cfg.stagger = rounds > 0;
    std::vector<float> blob;
    if (cfg.evaluator == C6_EVAL_NET) {      // net:FILE:CHANNELS:BLOCKS -- a weight blob as alphazero_connect6_amd.save_blob / c6_trainer_export_weights writes it
        char file[512]; int ch = 0, bl = 0;
        if (std::sscanf(ev.c_str(), "net:%511[^:]:%d:%d", file, &ch, &bl) != 3) { std::fprintf(stderr, "evaluator: net:FILE:CHANNELS:BLOCKS\n"); return 2; }
        cfg.net_channels = ch; cfg.net_blocks = bl;
        FILE *f = std::fopen(file, "rb");
        if (!f) { std::perror(file); return 1; }
        std::fseek(f, 0, SEEK_END); const long bytes = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        blob.resize((size_t)bytes / 4);
        if (std::fread(blob.data(), 4, blob.size(), f) != blob.size()) { std::fprintf(stderr, "short read of %s\n", file); return 1; }
        std::fclose(f);
    }
    c6_group *g = nullptr;
    if (c6_group_create(&cfg, dev.empty() ? nullptr : dev.data(), (int)dev.size(), &g) != C6_OK) die("c6_group_create");
    const int members = c6_group_size(g);
    if (!blob.empty() && c6_group_load_weights(g, blob.data(), blob.size()) != C6_OK) die("c6_group_load_weights");
    const size_t stride = c6_tuple_stride(n), cells = (size_t)n * n;
    const size_t cap = rounds > 0 ? 8192 : (size_t)games * cells;
    if (c6_group_enable_tuples(g, cap) != C6_OK) die("c6_group_enable_tuples");

    std::vector<uint32_t> counts(members);
    size_t padded = 0;
    uint64_t dropped = 0, dropped_all = 0, pooled_records = 0, pools = 0, payload_pools = 0;
    unsigned long long sum = 0, xr = 0;      // order-independent checksums of the pooled records (member 0's copy), 8 bytes at a time
    std::vector<unsigned char> host;
    auto digest = [&]() {
        host.resize((size_t)members * padded * stride + 8);
        size_t got = 0;
        if (c6_group_fetch_pooled(g, 0, (size_t)members * padded, &got, host.data()) != C6_OK) die("c6_group_fetch_pooled");
        for (size_t i = 0; i < got; i++) {
            unsigned long long h = 1469598103934665603ull;      // FNV-1a per record, then summed / xored over records: any order gives the same digest
            for (size_t k = 0; k < stride; k++) { h ^= host[i * stride + k]; h *= 1099511628211ull; }
            sum += h; xr ^= h;
        }
        pooled_records += got;
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (rounds > 0) {
        const int k = pool_every > 0 ? pool_every : 64;
        for (long done = 0; done < rounds; done += k) {
            const int c = (int)std::min<long>(k, rounds - done);
            if (done > 0 && c6_group_pool_begin(g) != C6_OK) die("c6_group_pool_begin");      // retire what the previous interval wrote ...
            if (c6_group_step(g, c) != C6_OK) die("c6_group_step");                          // ... enqueue this interval's rounds ...
            if (done > 0) {                                                                    // ... and pool behind them
                if (c6_group_pool_end(g, counts.data(), &padded, &dropped) != C6_OK) die("c6_group_pool_end");
                pools++; payload_pools += padded > 0; dropped_all += dropped;
                if (padded) digest();
            }
        }
        if (c6_group_sync(g) != C6_OK) die("c6_group_sync");
    } else {
        const int rc = c6_group_run(g, 0);
        if (rc != C6_OK && rc != C6_ERR_OVERFLOW) die("c6_group_run");
    }
    const double dt_play = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (c6_group_pool_tuples(g, counts.data(), &padded, &dropped) != C6_OK) die("c6_group_pool_tuples");
    pools++; payload_pools += padded > 0; dropped_all += dropped;
    if (padded) digest();
    // every member's copy of the last pool must be the same bytes
    int copies_equal = 1;
    if (padded && members > 1) {
        std::vector<unsigned char> a((size_t)members * padded * stride), b(a.size());
        size_t ga = 0, gb = 0;
        c6_group_fetch_pooled(g, 0, (size_t)members * padded, &ga, a.data());
        for (int r = 1; r < members; r++) {
            c6_group_fetch_pooled(g, r, (size_t)members * padded, &gb, b.data());
            if (ga != gb || std::memcmp(a.data(), b.data(), ga * stride) != 0) copies_equal = 0;
        }
    }
    c6_stats st;
    if (c6_group_stats(g, &st) != C6_OK) die("c6_group_stats");
    std::vector<int> winners(std::max(4 * games, 1024)), lens(winners.size());
    std::vector<uint32_t> uids(winners.size());
    int nfin = 0;
    if (c6_group_fetch_games(g, (int)winners.size(), &nfin, winners.data(), lens.data(), uids.data(), nullptr) != C6_OK) die("c6_group_fetch_games");
    unsigned long long uid_sum = 0, len_sum = 0;
    long wins[3] = {0, 0, 0};
    for (int i = 0; i < nfin; i++) { uid_sum += uids[i]; len_sum += (unsigned long long)lens[i]; wins[winners[i] + 1]++; }
    std::printf("{\"members\": %d, \"uses_rccl\": %d, \"board\": %d, \"games\": %d, \"sims\": %d, \"evaluator\": \"%s\", \"rounds\": %llu, \"seconds\": %.3f, "
                "\"simulations\": %llu, \"sims_per_s\": %.0f, \"stones\": %llu, \"games_finished\": %llu, \"fetched_games\": %d, \"uid_sum\": %llu, \"len_sum\": %llu, "
                "\"black_wins\": %ld, \"draws\": %ld, \"white_wins\": %ld, \"tuples\": %llu, \"pooled_records\": %llu, \"pools\": %llu, \"payload_pools\": %llu, "
                "\"dropped\": %llu, \"digest_sum\": %llu, \"digest_xor\": %llu, \"copies_equal\": %d, \"overflow_games\": %u}\n",
                members, c6_group_uses_rccl(g), n, games, sims, ev.c_str(), (unsigned long long)st.rounds, dt_play, (unsigned long long)st.simulations,
                st.simulations / dt_play, (unsigned long long)st.stones, (unsigned long long)st.games_finished, nfin, uid_sum, len_sum, wins[0], wins[1], wins[2],
                (unsigned long long)st.tuples, (unsigned long long)pooled_records, (unsigned long long)pools, (unsigned long long)payload_pools,
                (unsigned long long)dropped_all, sum, xr, copies_equal, st.overflow_games);
    c6_group_destroy(g);
    return 0;
}

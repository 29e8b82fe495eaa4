// The reference's outer loop -- AlphaZero/cpp_weighted/main.cpp:138-168: self_play -> buffer.push_game -> sample / loss / backward / step -> reload -- as a
// compiled C++ host over EVERY GPU of the node, nothing but libconnect6_amd's C ABI (include/connect6_amd.h parts 2, 3, 4 and 6), no Python, no launcher:
//     epoch:  c6_group_run               self-play of GAMES games sharded over the devices (resident net, the epoch's weights on every member)
//             c6_group_pool_tuples       ncclAllGather of the (state, pi, z) tuples: every member holds all of them
//             c6_group_push_pooled       member 0's copy -> the device-resident replay buffer (ReplayBuffer::push_game, ReplayBuffer.hpp:22-29)
//             c6_trainer_step_replay x K sample + loss + backward + SGD step on member 0's device (model.hpp:97-108, main.cpp:151-157)
//             c6_trainer_export_weights -> next epoch's c6_group_load_weights
//   g++ -O2 -std=c++17 -I include tools/cpp/group_loop.cpp -L alphazero-connect6_amd -lconnect6_amd -Wl,-rpath,'$ORIGIN/../../alphazero-connect6_amd' -o tools/cpp/_group_loop
//   tools/cpp/_group_loop DEVICES BOARD CHANNELS BLOCKS GAMES SIMS EPOCHS TRAIN_STEPS MINI_BATCH      (DEVICES: "all" | "0,1" | "0,0" = a rehearsal on one GPU)
// One JSON line per epoch.
#include "connect6_amd.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static void die(const char *what, const char *msg) { std::fprintf(stderr, "group_loop: %s: %s\n", what, msg); std::exit(1); }

// random-init weights in the blob layout of c6_engine_load_weights (header, above c6_net_blob_floats): conv / linear ~ U(-1/sqrt(fan_in), ..), BatchNorm
// gamma 1, beta 0, running mean 0, running variance 1 -- PyTorch's defaults, what `WeightedPolicy model(10)` starts from (main.cpp:12)
static std::vector<float> random_blob(int n, int C, int blocks, unsigned seed) {
    std::mt19937 rng(seed);
    std::vector<float> b;
    const int cells = n * n;
    auto uni = [&](size_t count, int fan_in) {
        std::uniform_real_distribution<float> d(-1.f / std::sqrt((float)fan_in), 1.f / std::sqrt((float)fan_in));
        for (size_t i = 0; i < count; i++) b.push_back(d(rng));
    };
    auto bn = [&](int k) {
        b.insert(b.end(), k, 1.f); b.insert(b.end(), k, 0.f); b.insert(b.end(), k, 0.f); b.insert(b.end(), k, 1.f);
    };
    uni((size_t)C * 2 * 9, 18); uni(C, 18); bn(C);
    for (int l = 0; l < 2 * (blocks - 1); l++) { uni((size_t)C * C * 9, 9 * C); uni(C, 9 * C); bn(C); }
    uni((size_t)2 * C, C); uni(2, C); bn(2);
    uni((size_t)cells * 2 * cells, 2 * cells); uni(cells, 2 * cells);
    uni(C, C); uni(1, C); bn(cells);
    uni((size_t)256 * cells, cells); uni(256, cells); uni(256, 256); uni(1, 256);
    return b;
}

int main(int argc, char **argv) {
    if (argc < 10) { std::fprintf(stderr, "usage: %s DEVICES BOARD CHANNELS BLOCKS GAMES SIMS EPOCHS TRAIN_STEPS MINI_BATCH\n", argv[0]); return 2; }
    std::vector<int> dev;
    if (std::strcmp(argv[1], "all") != 0)
        for (const char *p = argv[1]; *p;) { char *e; dev.push_back((int)std::strtol(p, &e, 10)); p = *e == ',' ? e + 1 : e; }
    const int n = std::atoi(argv[2]), C = std::atoi(argv[3]), blocks = std::atoi(argv[4]), games = std::atoi(argv[5]), sims = std::atoi(argv[6]);
    const int epochs = std::atoi(argv[7]), train_steps = std::atoi(argv[8]), batch = std::atoi(argv[9]);
    std::vector<float> blob = random_blob(n, C, blocks, 1);
    if (blob.size() != c6_net_blob_floats(n, C, blocks)) die("blob", "layout does not match c6_net_blob_floats");
    const int dev0 = dev.empty() ? 0 : dev[0];
    c6_replay *rb = nullptr;
    if (c6_replay_create_on_device(n, 1 << 20, dev0, &rb) != C6_OK) die("c6_replay_create_on_device", c6_last_error());
    c6_trainer *tr = nullptr;
    if (c6_trainer_create(n, C, blocks, batch, 1e-3f, 0.9f, dev0, &tr) != C6_OK) die("c6_trainer_create", c6_trainer_last_error());
    if (c6_trainer_load_weights(tr, blob.data(), blob.size()) != C6_OK) die("c6_trainer_load_weights", c6_trainer_last_error());
    uint32_t draw = 0;
    for (int ep = 0; ep < epochs; ep++) {
        const auto t0 = std::chrono::steady_clock::now();
        c6_config cfg;
        c6_default_config(&cfg);
        cfg.board_size = n; cfg.num_games = games; cfg.num_simulation = sims; cfg.seed = 1000 + ep;
        cfg.evaluator = C6_EVAL_NET; cfg.net_channels = C; cfg.net_blocks = blocks;
        c6_group *g = nullptr;
        if (c6_group_create(&cfg, dev.empty() ? nullptr : dev.data(), (int)dev.size(), &g) != C6_OK) die("c6_group_create", c6_last_error());
        const int members = c6_group_size(g);
        if (c6_group_load_weights(g, blob.data(), blob.size()) != C6_OK) die("c6_group_load_weights", c6_last_error());
        if (c6_group_enable_tuples(g, (size_t)(games / members + 1) * n * n) != C6_OK) die("c6_group_enable_tuples", c6_last_error());
        const int rc = c6_group_run(g, 0);
        if (rc != C6_OK && rc != C6_ERR_OVERFLOW) die("c6_group_run", c6_last_error());
        std::vector<uint32_t> counts(members);
        size_t padded = 0;
        uint64_t dropped = 0;
        if (c6_group_pool_tuples(g, counts.data(), &padded, &dropped) != C6_OK) die("c6_group_pool_tuples", c6_last_error());
        if (c6_group_push_pooled(g, 0, rb) != C6_OK) die("c6_group_push_pooled", c6_last_error());
        c6_stats st;
        c6_group_stats(g, &st);
        const int rccl = c6_group_uses_rccl(g);
        c6_group_destroy(g);
        const auto t1 = std::chrono::steady_clock::now();
        float first = 0.f, last = 0.f;
        for (int k = 0; k < train_steps; k++) {
            float loss = 0.f;
            if (c6_trainer_step_replay(tr, rb, 7, draw++, &loss) != C6_OK) die("c6_trainer_step_replay", c6_trainer_last_error());
            if (k == 0) first = loss;
            last = loss;
        }
        if (c6_trainer_export_weights(tr, blob.data(), blob.size()) != C6_OK) die("c6_trainer_export_weights", c6_trainer_last_error());
        const auto t2 = std::chrono::steady_clock::now();
        size_t num_data = 0, tail = 0;
        c6_replay_size(rb, &num_data, &tail);
        unsigned long long pooled = 0;
        for (uint32_t c : counts) pooled += c;
        std::printf("{\"epoch\": %d, \"members\": %d, \"uses_rccl\": %d, \"games_finished\": %llu, \"stones\": %llu, \"tuples\": %llu, \"pooled\": %llu, \"dropped\": %llu, "
                    "\"buffer\": %zu, \"self_play_s\": %.2f, \"train_s\": %.2f, \"loss_first\": %.4f, \"loss_last\": %.4f}\n",
                    ep, members, rccl, (unsigned long long)st.games_finished, (unsigned long long)st.stones, (unsigned long long)st.tuples, pooled,
                    (unsigned long long)dropped, num_data, std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), first, last);
        std::fflush(stdout);
    }
    c6_trainer_destroy(tr);
    c6_replay_destroy(rb);
    return 0;
}

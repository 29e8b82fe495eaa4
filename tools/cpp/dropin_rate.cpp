// Host-callback (PCIe-inclusive) rate of the drop-in boundary from C++: cpp_self_play with G concurrent 15x15 games and a trivial evaluator callback of the
// reference's shape (Connect6/cppbind/headers/connect6.hpp:14: void(int player, float* values, float* policies, int len)) -- no Python in the loop.
//   g++ -O2 -std=c++17 -I include tools/cpp/dropin_rate.cpp -L alphazero-connect6_amd -lconnect6_amd -Wl,-rpath,'$ORIGIN/../../alphazero-connect6_amd' -o tools/cpp/_dropin_rate
//   tools/cpp/_dropin_rate [games = 256] [sims = 50]
#include "connect6_amd.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
static long g_calls = 0, g_boards = 0;
static void callback(int, float *values, float *policies, int len) {
    g_calls++; g_boards += len;
    for (int i = 0; i < len; i++) values[i] = 0.f;
    const float u = 1.f / C6_REF_BOARD_CAPACITY;
    for (long i = 0; i < (long)len * C6_REF_BOARD_CAPACITY; i++) policies[i] = u;
}
static c6_path *alloc_path(int n) { return new c6_path[n]; }
static c6_play_result *alloc_result(int n) { return new c6_play_result[n]; }
int main(int argc, char **argv) {
    const int games = argc > 1 ? atoi(argv[1]) : 256, sims = argc > 2 ? atoi(argv[2]) : 50;
    const auto t0 = std::chrono::steady_clock::now();
    c6_vec v = cpp_self_play(callback, alloc_path, alloc_result, sims, 0.25f, 0.03, 1.0f, false, games);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long stones = 0;
    for (int i = 0; i < v.len; i++) { stones += v.vec[i].len; delete[] v.vec[i].paths; }
    delete[] v.vec;
    printf("{\"games\": %d, \"sims_per_move\": %d, \"stones\": %ld, \"wall_s\": %.2f, \"sims_per_s\": %.0f, \"boards_per_s\": %.0f, \"callback_calls\": %ld, \"bytes_over_pcie_per_s\": %.0f}\n",
           games, sims, stones, dt, stones * (double)sims / dt, g_boards / dt, g_calls, g_boards * 225.0 * 4 * 2 / dt);
    return 0;
}

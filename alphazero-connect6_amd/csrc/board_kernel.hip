// board_kernel.hip -- K1: place stone / six-in-a-row / legal mask on structure-of-arrays boards.
//
// Replaces, for G games at once:
//   Game::set        Connect6/libconnect6/src/game/game_impl/mod.rs:148-169
//   is_game_end      game_impl/mod.rs:249-252  ->  search, search_winner/mod.rs:201-246
// Layout: boards[cell * Gpad + game], int8 in {-1, 0, +1}; Gpad % 4 == 0.  One lane owns four
// consecutive games and loads their cells with one 4-byte load, so a wave-instruction moves 256
// contiguous bytes.  HBM-bound: N*N bytes read + 1 stone + status written per board-step.
//
// The winner scan is the bit-parallel form of the reference's row-major DP: row r of a colour is
// a bit mask; with a sliding window of six rows,
//   E_right = m & m<<1 & .. & m<<5          E_down  = AND_k row[r-k]
//   E_rdown = AND_k row[r-k] << k           E_ldown = AND_k row[r-k] >> k
// are exactly the cells of row r at which the reference's run counter of that direction reaches
// >= 6 (search_winner/mod.rs:206-222).  The first row with a non-empty E (black | white) and its
// lowest column give the first such cell in row-major order, which is the reference's answer --
// overlines and boards with several sixes included.
#include "c6_device.h"
#include "c6_engine.h"

#include "../../include/connect6_amd.h"

template <int N, bool STEP>
__global__ __launch_bounds__(256) void c6_board_kernel(int Gpad, int8_t *__restrict__ boards, int8_t *__restrict__ turn,
                                                      int8_t *__restrict__ remain, const int32_t *__restrict__ pos,
                                                      int8_t *__restrict__ winner, int8_t *__restrict__ err) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int g0 = q * 4;
    if (g0 >= Gpad) return;

    int pr[4], pc[4], mover[4];
    bool act[4], placed[4];
    int e[4] = {0, 0, 0, 0};
    bool any = !STEP;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        act[k] = !STEP;
        placed[k] = false;
        pr[k] = -1;
        pc[k] = 0;
        mover[k] = 0;
        if (STEP) {
            int p = pos[g0 + k];
            if (p >= N * N) e[k] = C6_ERR_INVALID_POSITION;    // game_impl/mod.rs:151-153
            else if (p >= 0) { act[k] = true; any = true; pr[k] = p / N; pc[k] = p % N; mover[k] = turn[g0 + k]; }
        }
    }
    if (STEP && !any) {
#pragma unroll
        for (int k = 0; k < 4; k++) if (e[k]) err[g0 + k] = (int8_t)e[k];
        return;
    }

    uint32_t wb[4][6], ww[4][6];   // sliding window, statically indexed after full unrolling
    int win[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < 6; i++) { wb[k][i] = 0; ww[k][i] = 0; }

#pragma unroll
    for (int r = 0; r < N; r++) {
        uint32_t mb[4] = {0, 0, 0, 0}, mw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < N; c++) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(boards + (size_t)(r * N + c) * Gpad + g0);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t b = (v >> (8 * k)) & 0xffu;
                mb[k] |= (uint32_t)(b == 0xffu) << c;
                mw[k] |= (uint32_t)(b == 0x01u) << c;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (STEP && act[k] && r == pr[k]) {
                if (((mb[k] | mw[k]) >> pc[k]) & 1u) {          // game_impl/mod.rs:155-157
                    e[k] = C6_ERR_ALREADY_SET;
                    act[k] = false;
                } else {
                    if (mover[k] < 0) mb[k] |= 1u << pc[k]; else mw[k] |= 1u << pc[k];
                    boards[(size_t)(pr[k] * N + pc[k]) * Gpad + g0 + k] = (int8_t)mover[k];   // :158
                    placed[k] = true;
                }
            }
            wb[k][r % 6] = mb[k];
            ww[k][r % 6] = mw[k];
            if (win[k] == 0 && act[k]) {
                uint32_t Eb, Ew;
                {
                    const uint32_t m = mb[k];
                    Eb = m & (m << 1) & (m << 2) & (m << 3) & (m << 4) & (m << 5);
                    const uint32_t n = mw[k];
                    Ew = n & (n << 1) & (n << 2) & (n << 3) & (n << 4) & (n << 5);
                }
                if (r >= 5) {
                    uint32_t vb = ~0u, rdb = ~0u, ldb = ~0u, vw = ~0u, rdw = ~0u, ldw = ~0u;
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        const uint32_t rb = wb[k][(r - i) % 6], rw = ww[k][(r - i) % 6];
                        vb &= rb; rdb &= rb << i; ldb &= rb >> i;
                        vw &= rw; rdw &= rw << i; ldw &= rw >> i;
                    }
                    Eb |= vb | rdb | ldb;
                    Ew |= vw | rdw | ldw;
                }
                const uint32_t E = Eb | Ew;
                if (E) {
                    const int c = __ffs((int)E) - 1;
                    win[k] = ((Eb >> c) & 1u) ? C6_BLACK : C6_WHITE;
                }
            }
        }
    }

#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (STEP) {
            err[g0 + k] = (int8_t)e[k];
            if (placed[k]) {
                winner[g0 + k] = (int8_t)win[k];
                int rem = remain[g0 + k] - 1;                 // game_impl/mod.rs:160
                if (rem <= 0) {                               // :164-167
                    rem = 2;
                    turn[g0 + k] = (int8_t)(-mover[k]);
                }
                remain[g0 + k] = (int8_t)rem;
            }
        } else {
            winner[g0 + k] = (int8_t)win[k];
        }
    }
}

template <int N>
__global__ __launch_bounds__(256) void c6_legal_mask_kernel(int Gpad, const int8_t *__restrict__ boards, uint8_t *__restrict__ mask) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int g0 = q * 4;
    if (g0 >= Gpad) return;
    constexpr int NB = (N * N + 7) / 8;
#pragma unroll 1
    for (int by = 0; by < NB; by++) {
        uint32_t out = 0;
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const int c = by * 8 + bit;
            if (c < N * N) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(boards + (size_t)c * Gpad + g0);
#pragma unroll
                for (int k = 0; k < 4; k++) out |= (uint32_t)(((v >> (8 * k)) & 0xffu) == 0u) << (8 * k + bit);
            }
        }
        *reinterpret_cast<uint32_t *>(mask + (size_t)by * Gpad + g0) = out;
    }
}

template <bool STEP>
static void launch_board(int N, int Gpad, int8_t *boards, int8_t *turn, int8_t *remain, const int32_t *pos, int8_t *winner,
                         int8_t *err, hipStream_t s) {
    const int quads = Gpad / 4, block = 256, grid = (quads + block - 1) / block;
    if (grid == 0) return;
    switch (N) {
    case 9: hipLaunchKernelGGL((c6_board_kernel<9, STEP>), dim3(grid), dim3(block), 0, s, Gpad, boards, turn, remain, pos, winner, err); break;
    case 15: hipLaunchKernelGGL((c6_board_kernel<15, STEP>), dim3(grid), dim3(block), 0, s, Gpad, boards, turn, remain, pos, winner, err); break;
    case 19: hipLaunchKernelGGL((c6_board_kernel<19, STEP>), dim3(grid), dim3(block), 0, s, Gpad, boards, turn, remain, pos, winner, err); break;
    default: break;
    }
}

void c6_launch_board_step(int N, int Gpad, int8_t *boards_soa, int8_t *turn, int8_t *remain, const int32_t *pos,
                          int8_t *winner, int8_t *err, hipStream_t s) {
    launch_board<true>(N, Gpad, boards_soa, turn, remain, pos, winner, err, s);
}

void c6_launch_search_winner(int N, int Gpad, const int8_t *boards_soa, int8_t *winner, hipStream_t s) {
    launch_board<false>(N, Gpad, const_cast<int8_t *>(boards_soa), nullptr, nullptr, nullptr, winner, nullptr, s);
}

void c6_launch_legal_mask(int N, int Gpad, const int8_t *boards_soa, uint8_t *mask, hipStream_t s) {
    const int quads = Gpad / 4, block = 256, grid = (quads + block - 1) / block;
    if (grid == 0) return;
    switch (N) {
    case 9: hipLaunchKernelGGL((c6_legal_mask_kernel<9>), dim3(grid), dim3(block), 0, s, Gpad, boards_soa, mask); break;
    case 15: hipLaunchKernelGGL((c6_legal_mask_kernel<15>), dim3(grid), dim3(block), 0, s, Gpad, boards_soa, mask); break;
    case 19: hipLaunchKernelGGL((c6_legal_mask_kernel<19>), dim3(grid), dim3(block), 0, s, Gpad, boards_soa, mask); break;
    default: break;
    }
}

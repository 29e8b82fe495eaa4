// board_kernel.hip -- K1: place stone / six-in-a-row / legal mask on structure-of-arrays boards.
//
// Replaces, for G games at once:
//   Game::set        Connect6/libconnect6/src/game/game_impl/mod.rs:148-169
//   is_game_end      game_impl/mod.rs:249-252  ->  search, search_winner/mod.rs:201-246
// Layout: boards[cell * Gpad + game], int8 in {-1, 0, +1}; Gpad % 4 == 0.  One lane owns four
// consecutive games and loads their cells with one 4-byte load, so a wave-instruction moves 256
// contiguous bytes.  HBM-bound: N*N bytes read + 1 stone + status written per board-step.
//
// Cell decode is SWAR over the four games of a dword (6 VALU per cell for 4 games): with M = 0x01010101,
//   black = (v >> 7) & M   (0xff has bit 7, 0x01 and 0x00 do not)      white = (v & M) ^ black
// and the one-bit-per-byte results are OR-ed, shifted by (column & 7), into byte-wise row accumulators (columns 0-7,
// 8-15, 16-18); a game's row mask is then byte k of the three accumulators (one v_perm + bfe/lshl_or).
//
// The winner scan is the bit-parallel form of the reference's row-major DP: row r of a colour is a bit mask m[r]; a run
// of six ending at a cell is found from shorter runs (pairs of pairs along a row, triples of triples across rows):
//   right:  h2 = m & m<<1            h4 = h2 & h2<<2                  E = h4 & h2<<4
//   down:   V3[r] = m[r] & m[r-1] & m[r-2]                            E = V3[r] & V3[r-3]
//   rdown:  D3[r] = m[r] & m[r-1]<<1 & m[r-2]<<2                      E = D3[r] & D3[r-3]<<3     (ldown: >> instead of <<)
// (rows above the board are empty masks, so no row guard is needed).  E holds exactly the cells of row r at which the
// reference's run counter of that direction reaches >= 6 (search_winner/mod.rs:206-222).  The first row with a non-empty
// E (black | white) and its lowest column give the first such cell in row-major order, which is the reference's answer --
// overlines and boards with several sixes included.  21 VALU per colour and row instead of the 46 of a re-evaluated
// six-row window.  c6_board_kernel_wide (window form, per-byte decode, plain 64-bit global loads) is the fallback the launcher
// takes for board arrays beyond the 32-bit row descriptors of c6_board_kernel (N * Gpad > 4 GiB: more than 226 M resident
// 19x19 games); C6_BOARD_WIDE=1 forces it so that the tests can reach it with small arrays (tests/test_gpu_variants.py).
#include <cstdlib>

#include "c6_device.h"
#include "c6_engine.h"

#include "../../include/connect6_amd.h"

template <int N, bool STEP>
__global__ __launch_bounds__(256) void c6_board_kernel_wide(int Gpad, int8_t *__restrict__ boards, int8_t *__restrict__ turn,
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

template <int N, bool STEP>
__global__ __launch_bounds__(256) void c6_board_kernel(int Gpad, int8_t *__restrict__ boards, int8_t *__restrict__ turn,
                                                      int8_t *__restrict__ remain, const int32_t *__restrict__ pos,
                                                      int8_t *__restrict__ winner, int8_t *__restrict__ err) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int g0 = q * 4;
    if (g0 >= Gpad) return;

    // Game::set of the four games, branch-free inside the scan: row pr[k] gets the new stone OR-ed into the mover's mask and
    // its previous occupancy recorded (a stone on an occupied cell is refused afterwards: nothing of that game is written).
    int e[4] = {0, 0, 0, 0}, mover[4] = {0, 0, 0, 0}, cellidx[4] = {0, 0, 0, 0};
    bool placed[4] = {!STEP, !STEP, !STEP, !STEP};     // search-only launch: every game is scanned and written
    uint32_t rowsel[4] = {0, 0, 0, 0}, stone[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, occ[4] = {0, 0, 0, 0};
    if (STEP) {
        const int4 p4 = *reinterpret_cast<const int4 *>(pos + g0);
        const uint32_t t4 = *reinterpret_cast<const uint32_t *>(turn + g0);
        const int pp[4] = {p4.x, p4.y, p4.z, p4.w};
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p = pp[k];
            const bool inr = p >= 0 && p < N * N;
            if (p >= N * N) e[k] = C6_ERR_INVALID_POSITION;    // game_impl/mod.rs:151-153
            any |= inr;
            cellidx[k] = inr ? p : 0;
            placed[k] = inr;
            mover[k] = (int)(int8_t)(t4 >> (8 * k));
            const int pr = cellidx[k] / N, pc = cellidx[k] % N;
            rowsel[k] = inr ? 1u << pr : 0u;
            stone[0][k] = mover[k] < 0 ? 1u << pc : 0u;
            stone[1][k] = mover[k] < 0 ? 0u : 1u << pc;
        }
        if (!any) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (e[k]) err[g0 + k] = (int8_t)e[k];
            return;
        }
    }

    // histories of the scan, [colour][game]: the two previous rows and, per column direction, the triples (rows r, r-1, r-2
    // all set along the direction) of the three previous rows [row % 3]; statically indexed after full unrolling
    uint32_t m1[2][4], m2[2][4], v3[2][4][3], d3[2][4][3], l3[2][4][3];
    uint32_t first[2][4];      // E (black, white) of the first row that has a six
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            m1[s][k] = 0;
            m2[s][k] = 0;
            first[s][k] = 0;
#pragma unroll
            for (int i = 0; i < 3; i++) { v3[s][k][i] = 0; d3[s][k][i] = 0; l3[s][k][i] = 0; }
        }

    constexpr uint32_t M = 0x01010101u;
    constexpr int NA = (N + 7) / 8;
    // Rows are loaded PF rows ahead of the scan (19-38 dword loads = 5-10 KB in flight per wave; left to itself hipcc
    // issues them one by one in front of their first use and the kernel is bound by the latency of single loads).
    constexpr int PF = N >= 15 ? 1 : 3, NBUF = PF + 1;
    // buffer loads: row descriptor (uniform, SGPRs) + scalar cell offset c * Gpad + the lane's game offset -- one VGPR of
    // address for all N*N loads (as global loads hipcc re-associates them into N*N per-lane 64-bit addresses and spills)
    const uint32_t goff = (uint32_t)g0;
    uint32_t rowbuf[NBUF][N];
    auto load_row = [&](int r) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(boards + (size_t)(r * N) * (size_t)(uint32_t)Gpad), 0, N * Gpad, 0x00020000);
#pragma unroll
        for (int c = 0; c < N; c++) rowbuf[r % NBUF][c] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)goff, c * Gpad, 0);
    };
#pragma unroll
    for (int r = 0; r < PF && r < N; r++) load_row(r);
#pragma unroll
    for (int r = 0; r < N; r++) {
        if (r + PF < N) load_row(r + PF);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t ab[NA], aw[NA];
#pragma unroll
        for (int i = 0; i < NA; i++) { ab[i] = 0; aw[i] = 0; }
#pragma unroll
        for (int c = 0; c < N; c++) {
            const uint32_t v = rowbuf[r % NBUF][c];
            const uint32_t t = (v >> 7) & M, w = (v & M) ^ t;
            ab[c >> 3] |= t << (c & 7);
            aw[c >> 3] |= w << (c & 7);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t m[2];
            // bytes {acc0.k, acc1.k, 0, 0}: v_perm_b32 (selector 0x0c = constant zero)
            const uint32_t sel = 0x0c0c0000u | ((4u + k) << 8) | (uint32_t)k;
            m[0] = __builtin_amdgcn_perm(ab[1], ab[0], sel);
            m[1] = __builtin_amdgcn_perm(aw[1], aw[0], sel);
            if (NA == 3) {
                m[0] |= ((ab[NA - 1] >> (8 * k)) & 0xffu) << 16;
                m[1] |= ((aw[NA - 1] >> (8 * k)) & 0xffu) << 16;
            }
            if (STEP) {      // the new stone (game_impl/mod.rs:158): all ones in row pr of a game that places, else 0
                const uint32_t here = (uint32_t)((int32_t)(rowsel[k] << (31 - r)) >> 31);      // v_bfe_i32
                occ[k] |= here & (m[0] | m[1]);
                m[0] |= here & stone[0][k];
                m[1] |= here & stone[1][k];
            }
            uint32_t E[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const uint32_t x = m[s];
                const uint32_t h2 = x & (x << 1), h4 = h2 & (h2 << 2);
                uint32_t Es = h4 & (h2 << 4);
                const uint32_t a = m1[s][k], b = m2[s][k];
                const uint32_t nv = x & a & b, nd = x & (a << 1) & (b << 2), nl = x & (a >> 1) & (b >> 2);
                if (r >= 5) {      // rows above the board are empty: E of the column directions is 0 before row 5
                    // the triples of row r-3 sit at r % 3, the slot this row overwrites afterwards
                    Es |= nv & v3[s][k][r % 3];
                    Es |= nd & (d3[s][k][r % 3] << 3);
                    Es |= nl & (l3[s][k][r % 3] >> 3);
                }
                v3[s][k][r % 3] = nv;
                d3[s][k][r % 3] = nd;
                l3[s][k][r % 3] = nl;
                m2[s][k] = a;
                m1[s][k] = x;
                E[s] = Es;
            }
            const bool none = (first[0][k] | first[1][k]) == 0u;
            first[0][k] = none ? E[0] : first[0][k];
            first[1][k] = none ? E[1] : first[1][k];
            // opaque: otherwise hipcc sinks the whole scan of game k into the `if (placed[k])` tail, keeps all 2N row masks
            // alive for it and runs it there with no loads in flight
            asm volatile("" : "+v"(first[0][k]), "+v"(first[1][k]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // results: one dword per status array and lane (its four games), merged with the previous bytes of games that do not
    // place; the stone itself is the only scattered store (one byte in a random plane per game)
    uint32_t w4 = 0, e4 = 0, keep = 0;      // keep: byte mask of the games whose winner / remain / turn stay as they are
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t Ea = first[0][k] | first[1][k];
        const int win = Ea == 0u ? 0 : ((first[0][k] & (Ea & (0u - Ea))) ? C6_BLACK : C6_WHITE);   // lowest column of that row
        if (STEP && placed[k] && (occ[k] & (stone[0][k] | stone[1][k]))) {      // game_impl/mod.rs:155-157
            e[k] = C6_ERR_ALREADY_SET;
            placed[k] = false;
        }
        w4 |= (uint32_t)(uint8_t)win << (8 * k);
        e4 |= (uint32_t)(uint8_t)e[k] << (8 * k);
        if (!placed[k]) keep |= 0xffu << (8 * k);
    }
    if (!STEP) {
        *reinterpret_cast<uint32_t *>(winner + g0) = w4;
        return;
    }
    *reinterpret_cast<uint32_t *>(err + g0) = e4;
    if (keep == 0xffffffffu) return;
    const uint32_t t4 = *reinterpret_cast<const uint32_t *>(turn + g0), r4 = *reinterpret_cast<const uint32_t *>(remain + g0);
    const uint32_t ow4 = *reinterpret_cast<const uint32_t *>(winner + g0);
    uint32_t nt4 = t4, nr4 = r4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (placed[k]) {
#ifndef C6_AB_BOARD_NO_STONE      // ablation builds only (tools/ab_build.sh): the board-step without its one scattered store
            boards[(size_t)cellidx[k] * Gpad + g0 + k] = (int8_t)mover[k];   // game_impl/mod.rs:158
#endif
            int rem = (int)(int8_t)(r4 >> (8 * k)) - 1;                      // game_impl/mod.rs:160
            if (rem <= 0) {                                                  // :164-167
                rem = 2;
                nt4 = (nt4 & ~(0xffu << (8 * k))) | ((uint32_t)(uint8_t)(-mover[k]) << (8 * k));
            }
            nr4 = (nr4 & ~(0xffu << (8 * k))) | ((uint32_t)(uint8_t)rem << (8 * k));
        }
    }
#ifdef C6_AB_BOARD_NO_STATUS      // ablation builds only: one status dword instead of three
    *reinterpret_cast<uint32_t *>(winner + g0) = ((ow4 & keep) | (w4 & ~keep)) ^ nr4 ^ nt4;
#else
    *reinterpret_cast<uint32_t *>(winner + g0) = (ow4 & keep) | (w4 & ~keep);
    *reinterpret_cast<uint32_t *>(remain + g0) = nr4;
    if (nt4 != t4) *reinterpret_cast<uint32_t *>(turn + g0) = nt4;
#endif
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
                out |= (~v & 0x01010101u) << bit;      // SWAR over the four games: bit 0 of a byte is set iff the cell is occupied
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
    static const bool v1env = [] { const char *e = getenv("C6_BOARD_WIDE"); return e && e[0] == '1'; }();   // test hook
    // the row descriptors of c6_board_kernel hold N * Gpad bytes in 32 bits (Gpad < 226 M games at N = 19: an 81 GB board
    // array); anything larger runs on the global-load kernel
    const bool v1 = v1env || (size_t)N * (size_t)Gpad > 0xffffffffull;
#define C6_BOARD_CASE(NN)                                                                                                        \
    case NN:                                                                                                                      \
        if (v1) hipLaunchKernelGGL((c6_board_kernel_wide<NN, STEP>), dim3(grid), dim3(block), 0, s, Gpad, boards, turn, remain, pos, winner, err); \
        else hipLaunchKernelGGL((c6_board_kernel<NN, STEP>), dim3(grid), dim3(block), 0, s, Gpad, boards, turn, remain, pos, winner, err);      \
        break;
    switch (N) {
        C6_BOARD_CASE(9)
        C6_BOARD_CASE(15)
        C6_BOARD_CASE(19)
    default: break;
    }
#undef C6_BOARD_CASE
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

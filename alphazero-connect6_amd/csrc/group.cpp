// group.cpp -- several GPUs behind the C ABI (include/connect6_amd.h part 6).
//
// The reference's only parallelism is data-parallel games INSIDE the library: AsyncAgent::run hands `num` games to a pool of one
// worker per core and collects the results over an mpsc channel (agent/async_agent/mod.rs:100-137, entered from
// cppbind/libcppconnect6/src/lib.rs:110-122).  c6_group is that for the GPUs of a node: one engine + one host thread per device,
// games sharded by uid, nothing exchanged while playing, and the (state, pi, z) training tuples pooled over the members by
// ncclAllGather (RCCL over xGMI) -- the mpsc collection's counterpart.  One process drives all devices (ncclCommInitAll: one
// communicator per member, collectives issued inside ncclGroupStart / ncclGroupEnd), so a C++ or Rust host needs no launcher.
//
// librccl is opened with dlopen when a group is created: the library itself carries no RCCL dependency (engines, the drop-in
// boundary and the trainer load without it), and members that share a device pool through host memory (RCCL admits one rank per
// device) -- that form is what a one-GPU box can rehearse with devices = {0, 0}; with devices = {0} the RCCL calls themselves run.
#include "../../include/connect6_amd.h"
#include "c6_engine.h"

int c6_set_error(int code, const char *msg);      // engine.cpp: the message c6_last_error() returns on this thread

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {

int gfail(int code, const std::string &msg) { return c6_set_error(code, msg.c_str()); }

#define GHIP(x)                                                                                                \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) return gfail(C6_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));       \
    } while (0)

// The group's entry points select devices on the CALLING thread (HIP's current device is per thread): a host that uses HIP itself -- a libtorch trainer
// calling cpp_self_play -- must find its own current device again afterwards, whichever member was touched last.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

// ---- librccl, resolved at run time --------------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return false; }
#define SYM(field, name)                                                                \
        field = reinterpret_cast<decltype(field)>(dlsym(lib, name));                    \
        if (!field) { err = std::string("librccl lacks ") + name; dlclose(lib); lib = nullptr; return false; }
        SYM(CommInitAll, "ncclCommInitAll") SYM(CommDestroy, "ncclCommDestroy") SYM(AllGather, "ncclAllGather")
        SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
        SYM(GetVersion, "ncclGetVersion")
#undef SYM
        return true;
    }
};
Rccl &rccl() {
    static Rccl r;
    return r;
}
std::mutex g_rccl_mu;      // dlopen once; communicator creation / destruction one group at a time

struct Member {
    int device = 0;
    uint32_t first_uid = 0;
    int games = 0;
    c6_engine *eng = nullptr;
    hipStream_t side = nullptr;          // the exchange's stream: the engine's own stream keeps playing
    hipEvent_t retired = nullptr;        // recorded on the engine's stream when a buffer is retired
    hipEvent_t freed = nullptr;          // recorded on the side stream: the buffer about to be written again has been pooled
    uint8_t *buf[2] = {nullptr, nullptr};
    uint32_t *cnt[2] = {nullptr, nullptr};
    uint8_t *pooled = nullptr;           // [ndev][padded][stride]
    size_t pooled_bytes = 0;
    uint32_t *pooled_counts = nullptr;   // device copy of the last pool's counts[ndev] (c6_replay_push_dev reads them there)
    ncclComm_t comm = nullptr;
};

}      // namespace

struct c6_group {
    c6_config cfg{};
    std::vector<Member> m;
    bool distinct = false, use_rccl = false;
    size_t cap = 0, stride = 0;          // tuple records per buffer, bytes per record
    int cur = 0;                          // the buffer set the engines write now
    bool begun = false;                   // a _begin without its _end
    std::vector<uint32_t> last_counts;
    size_t last_padded = 0;
    std::mutex cb_mu;
};

namespace {

// ---- the evaluator callback of a group: the members' host threads reach the user's function through one trampoline ------------
// (c6_callback carries no user pointer, so the group a thread works for is a thread-local)
thread_local c6_group *t_group = nullptr;
struct CbState {
    c6_callback user = nullptr;
    bool concurrent = false;
};
std::mutex g_cb_reg_mu;
std::vector<std::pair<c6_group *, CbState>> g_cb_reg;
CbState cb_of(c6_group *g) {
    std::lock_guard<std::mutex> lk(g_cb_reg_mu);
    for (auto &p : g_cb_reg)
        if (p.first == g) return p.second;
    return CbState{};
}
void group_trampoline(int player, float *values, float *policies, int len) {
    c6_group *g = t_group;
    const CbState st = g ? cb_of(g) : CbState{};
    if (!st.user) return;      // (values stay zero, the boards stay in `policies`: never reached -- set_callback installs both together)
    if (st.concurrent) st.user(player, values, policies, len);
    else {
        std::lock_guard<std::mutex> lk(g->cb_mu);
        st.user(player, values, policies, len);
    }
}

// run f(member index) on one host thread per member (the calling thread takes member 0); collects the first failure's status and message
template <typename F> int for_members(c6_group *g, F &&f) {
    const int n = (int)g->m.size();
    std::vector<int> rc(n, C6_OK);
    std::vector<std::string> msg(n);
    auto body = [&](int r) {
        t_group = g;
        if (hipSetDevice(g->m[r].device) != hipSuccess) { rc[r] = C6_ERR_HIP; msg[r] = "hipSetDevice failed"; return; }
        rc[r] = f(r);
        if (rc[r] != C6_OK) msg[r] = c6_last_error();      // (thread-local: read on the thread that failed)
        t_group = nullptr;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n; r++) th.emplace_back(body, r);
    body(0);
    for (auto &t : th) t.join();
    for (int r = 0; r < n; r++)
        if (rc[r] != C6_OK) return gfail(rc[r], "member " + std::to_string(r) + " (device " + std::to_string(g->m[r].device) + "): " + msg[r]);
    return C6_OK;
}

size_t padded_records(size_t m, size_t cap) {      // as shard.TuplePool: whole quanta of 64 records travel
    return m == 0 ? 0 : std::min(cap, (m + 63) / 64 * 64);
}

void free_member_buffers(Member &mb) {
    hipSetDevice(mb.device);
    for (int i = 0; i < 2; i++) {
        if (mb.buf[i]) hipFree(mb.buf[i]);
        if (mb.cnt[i]) hipFree(mb.cnt[i]);
        mb.buf[i] = nullptr;
        mb.cnt[i] = nullptr;
    }
    if (mb.pooled) hipFree(mb.pooled);
    if (mb.pooled_counts) hipFree(mb.pooled_counts);
    mb.pooled = nullptr;
    mb.pooled_counts = nullptr;
    mb.pooled_bytes = 0;
}

}      // namespace

// RCCL: one communicator per member, all created by this process -- when the group first needs them (c6_group_enable_tuples): a group that only
// plays (cpp_self_play) never pays for ncclCommInitAll.  C6_GROUP_NO_RCCL=1 forces the host-staged exchange (A/B, debugging).
static void group_init_rccl(c6_group *g) {
    const int n = (int)g->m.size();
    const char *no = std::getenv("C6_GROUP_NO_RCCL");
    if (!g->distinct || (no && no[0] == '1') || g->use_rccl) return;
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    Rccl &R = rccl();
    if (!R.load()) {
        if (n > 1) std::fprintf(stderr, "[connect6_amd] c6_group: %s -- tuples are pooled through host memory\n", R.err.c_str());
        return;
    }
    std::vector<ncclComm_t> comms(n);
    std::vector<int> dev(n);
    for (int r = 0; r < n; r++) dev[r] = g->m[r].device;
    const ncclResult_t nr = R.CommInitAll(comms.data(), n, dev.data());
    if (nr != ncclSuccess) {
        std::fprintf(stderr, "[connect6_amd] c6_group: ncclCommInitAll over %d device(s): %s -- tuples are pooled through host memory\n", n, R.GetErrorString(nr));
        return;
    }
    for (int r = 0; r < n; r++) g->m[r].comm = comms[r];
    g->use_rccl = true;
}

extern "C" int c6_group_create(const c6_config *cfg, const int *devices, int ndev, c6_group **out) {
    DeviceGuard dg_;
    if (!cfg || !out) return gfail(C6_ERR_BAD_ARG, "null argument");
    const int visible = c6_device_count();
    if (visible < 1) return gfail(C6_ERR_NO_GPU, "no gfx950 device visible: libconnect6_amd has no CPU fallback");
    std::vector<int> dev;
    if (!devices || ndev <= 0)
        for (int d = 0; d < visible; d++) dev.push_back(d);
    else
        dev.assign(devices, devices + ndev);
    for (int d : dev)
        if (d < 0 || d >= visible) return gfail(C6_ERR_BAD_ARG, "c6_group_create: device " + std::to_string(d) + " is not visible (" + std::to_string(visible) + " device(s))");
    const int n = (int)dev.size();
    if (cfg->num_games < n) return gfail(C6_ERR_BAD_ARG, "c6_group_create: fewer games than members (every member plays at least one game)");
    c6_group *g = new c6_group();
    g->cfg = *cfg;
    g->m.resize(n);
    g->distinct = std::set<int>(dev.begin(), dev.end()).size() == dev.size();
    g->stride = c6_tuple_stride(cfg->board_size);
    // the shard plan of alphazero-connect6_amd/shard.py (shard_games + uid_plan): member r plays games [first, first + mine) of the call's G
    // games; a recycled slot continues with uid += G, so the union over the members is the game set of ONE engine with G slots
    const int per = cfg->num_games / n, rem = cfg->num_games % n;
    for (int r = 0; r < n; r++) {
        Member &mb = g->m[r];
        mb.device = dev[r];
        mb.games = per + (r < rem ? 1 : 0);
        mb.first_uid = cfg->game_uid_base + (uint32_t)(r * per + std::min(r, rem));
    }
    // the members' engines are built at the same time, each on its own host thread (pools of tens of GB per device: seconds each)
    const int rc = for_members(g, [&](int r) {
        Member &mb = g->m[r];
        c6_config c = *cfg;
        c.device = mb.device;
        c.num_games = mb.games;
        c.game_uid_base = mb.first_uid;
        c.game_uid_stride = (uint32_t)cfg->num_games;
        int rc_ = c6_engine_create(&c, &mb.eng);
        if (rc_ != C6_OK) return rc_;
        if (hipSetDevice(mb.device) != hipSuccess || hipStreamCreateWithFlags(&mb.side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&mb.retired, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&mb.freed, hipEventDisableTiming) != hipSuccess)
            return gfail(C6_ERR_HIP, "c6_group_create: side stream / event");
        return (int)C6_OK;
    });
    if (rc != C6_OK) {
        const std::string keep = c6_last_error();
        c6_group_destroy(g);
        return gfail(rc, keep);
    }
    *out = g;
    return C6_OK;
}

extern "C" void c6_group_destroy(c6_group *g) {
    DeviceGuard dg_;
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g_cb_reg_mu);
        g_cb_reg.erase(std::remove_if(g_cb_reg.begin(), g_cb_reg.end(), [&](const std::pair<c6_group *, CbState> &p) { return p.first == g; }), g_cb_reg.end());
    }
    for (Member &mb : g->m) {
        hipSetDevice(mb.device);
        if (mb.eng) c6_engine_sync(mb.eng);
        if (mb.side) hipStreamSynchronize(mb.side);
    }
    if (g->use_rccl) {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        for (Member &mb : g->m)
            if (mb.comm) rccl().CommDestroy(mb.comm);
    }
    for (Member &mb : g->m) {
        hipSetDevice(mb.device);
        if (mb.eng) c6_engine_destroy(mb.eng);
        free_member_buffers(mb);
        if (mb.side) hipStreamDestroy(mb.side);
        if (mb.retired) hipEventDestroy(mb.retired);
        if (mb.freed) hipEventDestroy(mb.freed);
    }
    delete g;
}

extern "C" int c6_group_size(const c6_group *g) { return g ? (int)g->m.size() : 0; }
extern "C" c6_engine *c6_group_engine(c6_group *g, int member) { return g && member >= 0 && member < (int)g->m.size() ? g->m[member].eng : nullptr; }
extern "C" int c6_group_member_info(const c6_group *g, int member, int *device, uint32_t *first_uid, int *games) {
    if (!g || member < 0 || member >= (int)g->m.size()) return gfail(C6_ERR_BAD_ARG, "c6_group_member_info: bad member");
    if (device) *device = g->m[member].device;
    if (first_uid) *first_uid = g->m[member].first_uid;
    if (games) *games = g->m[member].games;
    return C6_OK;
}
extern "C" int c6_group_uses_rccl(const c6_group *g) { return g && g->use_rccl ? 1 : 0; }

extern "C" int c6_group_load_weights(c6_group *g, const float *blob, size_t nfloats) {
    DeviceGuard dg_;
    if (!g || !blob) return gfail(C6_ERR_BAD_ARG, "null argument");
    return for_members(g, [&](int r) { return c6_engine_load_weights(g->m[r].eng, blob, nfloats); });
}

extern "C" int c6_group_set_callback(c6_group *g, c6_callback cb, int concurrent) {
    if (!g) return gfail(C6_ERR_BAD_ARG, "null group");
    {
        std::lock_guard<std::mutex> lk(g_cb_reg_mu);
        bool found = false;
        for (auto &p : g_cb_reg)
            if (p.first == g) { p.second = CbState{cb, concurrent != 0}; found = true; }
        if (!found) g_cb_reg.emplace_back(g, CbState{cb, concurrent != 0});
    }
    for (Member &mb : g->m) {
        int rc = c6_engine_set_callback(mb.eng, cb ? group_trampoline : nullptr);
        if (rc != C6_OK) return rc;
    }
    return C6_OK;
}

extern "C" int c6_group_step(c6_group *g, int rounds) {
    DeviceGuard dg_;
    if (!g) return gfail(C6_ERR_BAD_ARG, "null group");
    return for_members(g, [&](int r) { return c6_engine_step(g->m[r].eng, rounds); });
}

extern "C" int c6_group_run(c6_group *g, long max_rounds) {
    DeviceGuard dg_;
    if (!g) return gfail(C6_ERR_BAD_ARG, "null group");
    // every member to the end of its games, whatever the others do (a member with partial games -- C6_ERR_OVERFLOW -- does not stop the rest)
    return for_members(g, [&](int r) { return c6_engine_run(g->m[r].eng, max_rounds); });
}

extern "C" int c6_group_sync(c6_group *g) {
    DeviceGuard dg_;
    if (!g) return gfail(C6_ERR_BAD_ARG, "null group");
    for (Member &mb : g->m) {
        GHIP(hipSetDevice(mb.device));
        int rc = c6_engine_sync(mb.eng);
        if (rc != C6_OK) return rc;
        GHIP(hipStreamSynchronize(mb.side));
    }
    return C6_OK;
}

extern "C" int c6_group_stats(c6_group *g, c6_stats *out) {
    DeviceGuard dg_;
    if (!g || !out) return gfail(C6_ERR_BAD_ARG, "null argument");
    c6_stats s;
    std::memset(&s, 0, sizeof s);
    s.pool_pages_low = 0xffffffffu;
    for (Member &mb : g->m) {
        GHIP(hipSetDevice(mb.device));
        c6_stats t;
        int rc = c6_engine_stats(mb.eng, &t);
        if (rc != C6_OK) return rc;
        s.rounds = std::max(s.rounds, t.rounds);
        s.simulations += t.simulations; s.select_steps += t.select_steps; s.stones += t.stones; s.games_finished += t.games_finished;
        s.tuples += t.tuples; s.live_games += t.live_games; s.overflow_games += t.overflow_games; s.abandoned_games += t.abandoned_games;
        s.max_records_used = std::max(s.max_records_used, t.max_records_used);
        s.max_depth = std::max(s.max_depth, t.max_depth);
        s.records_per_game = std::max(s.records_per_game, t.records_per_game);
        s.pool_grows += t.pool_grows; s.pool_pages += t.pool_pages; s.pool_pages_free += t.pool_pages_free;
        s.pool_pages_low = std::min(s.pool_pages_low, t.pool_pages_low);
        s.cap_raises += t.cap_raises; s.parked_games += t.parked_games;
        s.max_pages_held = std::max(s.max_pages_held, t.max_pages_held);
        s.pool_bytes += t.pool_bytes;
    }
    *out = s;
    return C6_OK;
}

extern "C" int c6_group_fetch_games(c6_group *g, int cap, int *count, int *winners, int *lens, uint32_t *uids, int16_t *moves) {
    DeviceGuard dg_;
    if (!g || !count) return gfail(C6_ERR_BAD_ARG, "null argument");
    const size_t cells = (size_t)g->cfg.board_size * g->cfg.board_size;
    int got = 0;
    for (Member &mb : g->m) {
        if (got >= cap) break;
        GHIP(hipSetDevice(mb.device));
        int k = 0;
        int rc = c6_engine_fetch_games(mb.eng, cap - got, &k, winners ? winners + got : nullptr, lens ? lens + got : nullptr, uids ? uids + got : nullptr,
                                       moves ? moves + (size_t)got * cells : nullptr);
        if (rc != C6_OK) return rc;
        got += k;
    }
    *count = got;
    return C6_OK;
}

extern "C" int c6_group_enable_tuples(c6_group *g, size_t capacity_records) {
    DeviceGuard dg_;
    if (!g || capacity_records == 0) return gfail(C6_ERR_BAD_ARG, "c6_group_enable_tuples: null group / zero capacity");
    if (g->cap) return gfail(C6_ERR_BAD_ARG, "c6_group_enable_tuples: already enabled");
    const size_t n = g->m.size();
    group_init_rccl(g);
    for (Member &mb : g->m) {
        GHIP(hipSetDevice(mb.device));
        for (int i = 0; i < 2; i++) {
            GHIP(hipMalloc(&mb.buf[i], capacity_records * g->stride));
            GHIP(hipMemset(mb.buf[i], 0, capacity_records * g->stride));      // (a record's padding bytes are never written: zero, as in an engine-owned buffer)
            GHIP(hipMalloc(&mb.cnt[i], sizeof(uint32_t)));
            GHIP(hipMemset(mb.cnt[i], 0, sizeof(uint32_t)));
        }
        GHIP(hipMalloc(&mb.pooled_counts, n * sizeof(uint32_t)));
        GHIP(hipMemset(mb.pooled_counts, 0, n * sizeof(uint32_t)));
        int rc = c6_engine_set_tuple_buffer(mb.eng, mb.buf[0], capacity_records, mb.cnt[0]);
        if (rc != C6_OK) return rc;
    }
    g->cap = capacity_records;
    g->cur = 0;
    g->last_counts.assign(n, 0);
    return C6_OK;
}

extern "C" int c6_group_pool_begin(c6_group *g) {
    DeviceGuard dg_;
    if (!g) return gfail(C6_ERR_BAD_ARG, "null group");
    if (!g->cap) return gfail(C6_ERR_BAD_ARG, "c6_group_pool_begin: call c6_group_enable_tuples first");
    if (g->begun) return gfail(C6_ERR_BAD_ARG, "c6_group_pool_begin: the previous exchange has not been ended");
    const int nxt = g->cur ^ 1;
    for (Member &mb : g->m) {
        GHIP(hipSetDevice(mb.device));
        hipStream_t es = (hipStream_t)c6_engine_stream(mb.eng);
        // the engine may write `nxt` again only after its last pooling (side stream) is done
        GHIP(hipEventRecord(mb.freed, mb.side));
        GHIP(hipStreamWaitEvent(es, mb.freed, 0));
        int rc = c6_engine_swap_tuple_buffer(mb.eng, mb.buf[nxt], g->cap, mb.cnt[nxt]);      // no drain: the buffer travels with every launch
        if (rc != C6_OK) return rc;
        GHIP(hipEventRecord(mb.retired, es));                                                  // everything enqueued so far wrote `cur`
        GHIP(hipStreamWaitEvent(mb.side, mb.retired, 0));
    }
    g->cur = nxt;
    g->begun = true;
    return C6_OK;
}

extern "C" int c6_group_pool_end(c6_group *g, uint32_t *counts, size_t *padded, uint64_t *dropped) {
    DeviceGuard dg_;
    if (!g) return gfail(C6_ERR_BAD_ARG, "null group");
    if (!g->begun) return gfail(C6_ERR_BAD_ARG, "c6_group_pool_end without c6_group_pool_begin");
    const int old = g->cur ^ 1, n = (int)g->m.size();
    std::vector<uint32_t> raw(n, 0);
    for (int r = 0; r < n; r++) {      // the only host wait: rounds that were enqueued BEFORE _begin
        Member &mb = g->m[r];
        GHIP(hipSetDevice(mb.device));
        GHIP(hipMemcpyAsync(&raw[r], mb.cnt[old], sizeof(uint32_t), hipMemcpyDeviceToHost, mb.side));
    }
    for (Member &mb : g->m) { GHIP(hipSetDevice(mb.device)); GHIP(hipStreamSynchronize(mb.side)); }
    uint64_t lost = 0;
    size_t mx = 0;
    for (int r = 0; r < n; r++) {
        if (raw[r] > g->cap) { lost += raw[r] - g->cap; raw[r] = (uint32_t)g->cap; }      // the engine stops writing at cap and keeps counting
        mx = std::max<size_t>(mx, raw[r]);
    }
    const size_t pm = padded_records(mx, g->cap), seg = pm * g->stride;
    if (pm) {
        for (Member &mb : g->m) {
            if (mb.pooled_bytes < seg * n) {
                GHIP(hipSetDevice(mb.device));
                if (mb.pooled) GHIP(hipFree(mb.pooled));
                mb.pooled = nullptr;
                mb.pooled_bytes = 0;
                GHIP(hipMalloc(&mb.pooled, seg * n));
                mb.pooled_bytes = seg * n;
            }
        }
        if (g->use_rccl) {
            // one collective over all members: ncclAllGather of `pm` records from every member's retired buffer into every member's pooled array,
            // each on that member's side stream (RCCL moves the segments device to device over xGMI; a one-member group copies in place)
            Rccl &R = rccl();
            ncclResult_t nr = R.GroupStart();
            for (int r = 0; r < n && nr == ncclSuccess; r++) {
                Member &mb = g->m[r];
                nr = R.AllGather(mb.buf[old], mb.pooled, seg, ncclChar, mb.comm, mb.side);
            }
            const ncclResult_t ne = R.GroupEnd();
            if (nr == ncclSuccess) nr = ne;
            if (nr != ncclSuccess) return gfail(C6_ERR_HIP, std::string("ncclAllGather: ") + R.GetErrorString(nr));
        } else {
            // members sharing a device (or no librccl): the same result through page-able host memory
            std::vector<uint8_t> host(seg * n);
            for (int r = 0; r < n; r++) {
                Member &mb = g->m[r];
                GHIP(hipSetDevice(mb.device));
                GHIP(hipMemcpyAsync(host.data() + seg * r, mb.buf[old], seg, hipMemcpyDeviceToHost, mb.side));
            }
            for (Member &mb : g->m) { GHIP(hipSetDevice(mb.device)); GHIP(hipStreamSynchronize(mb.side)); }
            for (Member &mb : g->m) {
                GHIP(hipSetDevice(mb.device));
                GHIP(hipMemcpyAsync(mb.pooled, host.data(), seg * n, hipMemcpyHostToDevice, mb.side));
            }
            for (Member &mb : g->m) { GHIP(hipSetDevice(mb.device)); GHIP(hipStreamSynchronize(mb.side)); }
        }
    }
    for (Member &mb : g->m) {
        GHIP(hipSetDevice(mb.device));
        GHIP(hipMemcpyAsync(mb.pooled_counts, raw.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, mb.side));
        GHIP(hipMemsetAsync(mb.cnt[old], 0, sizeof(uint32_t), mb.side));
    }
    for (Member &mb : g->m) { GHIP(hipSetDevice(mb.device)); GHIP(hipStreamSynchronize(mb.side)); }
    g->last_counts = raw;
    g->last_padded = pm;
    g->begun = false;
    if (counts) std::copy(raw.begin(), raw.end(), counts);
    if (padded) *padded = pm;
    if (dropped) *dropped = lost;
    return C6_OK;
}

extern "C" int c6_group_pool_tuples(c6_group *g, uint32_t *counts, size_t *padded, uint64_t *dropped) {
    int rc = c6_group_pool_begin(g);
    return rc != C6_OK ? rc : c6_group_pool_end(g, counts, padded, dropped);
}

extern "C" const void *c6_group_pooled(const c6_group *g, int member) {
    return g && member >= 0 && member < (int)g->m.size() && g->last_padded ? g->m[member].pooled : nullptr;
}

extern "C" int c6_group_fetch_pooled(c6_group *g, int member, size_t cap, size_t *count, void *host_buf) {
    DeviceGuard dg_;
    if (!g || !count || member < 0 || member >= (int)g->m.size()) return gfail(C6_ERR_BAD_ARG, "c6_group_fetch_pooled: bad argument");
    Member &mb = g->m[member];
    GHIP(hipSetDevice(mb.device));
    size_t got = 0;
    const size_t seg = g->last_padded * g->stride;
    for (size_t r = 0; r < g->m.size() && got < cap; r++) {
        const size_t k = std::min<size_t>(g->last_counts[r], cap - got);
        if (k && host_buf) GHIP(hipMemcpy(static_cast<uint8_t *>(host_buf) + got * g->stride, mb.pooled + seg * r, k * g->stride, hipMemcpyDeviceToHost));
        got += k;
    }
    *count = got;
    return C6_OK;
}

extern "C" int c6_group_push_pooled(c6_group *g, int member, c6_replay *rp) {
    DeviceGuard dg_;
    if (!g || !rp || member < 0 || member >= (int)g->m.size()) return gfail(C6_ERR_BAD_ARG, "c6_group_push_pooled: bad argument");
    Member &mb = g->m[member];
    GHIP(hipSetDevice(mb.device));
    const size_t seg = g->last_padded * g->stride;
    for (size_t r = 0; r < g->m.size(); r++) {
        if (!g->last_counts[r]) continue;
        int rc = c6_replay_push_dev(rp, mb.pooled + seg * r, mb.pooled_counts + r, g->last_padded);
        if (rc != C6_OK) return gfail(rc, "c6_group_push_pooled: c6_replay_push_dev failed");
    }
    GHIP(hipDeviceSynchronize());
    return C6_OK;
}

// ---- cpp_self_play over several devices (abi_dropin.cpp) ----------------------------------------------------------------------------------
// C6_DEVICES = "all" | "0,1,2" (repeats allowed: rehearsal) | unset.  Unset: every visible device, but no more of them than the call's
// games fill at C6_GROUP_MIN_GAMES (default 64) games per device -- the reference's default of a handful of games stays on one GPU.
std::vector<int> c6_dropin_devices(int num_games) {
    const int visible = c6_device_count();
    std::vector<int> dev;
    const char *e = std::getenv("C6_DEVICES");
    if (e && *e && std::strcmp(e, "all") != 0) {
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long d = std::strtol(p, &end, 10);
            if (end == p) break;
            dev.push_back((int)d);
            p = *end == ',' ? end + 1 : end;
        }
    } else {
        int use = visible;
        if (!e) {
            const char *mg = std::getenv("C6_GROUP_MIN_GAMES");
            const int min_games = mg ? std::max(1, std::atoi(mg)) : 64;
            use = std::max(1, std::min(visible, num_games / min_games));
        }
        for (int d = 0; d < use; d++) dev.push_back(d);
    }
    while ((int)dev.size() > std::max(1, num_games)) dev.pop_back();      // at least one game per member
    if (dev.empty()) dev.push_back(0);
    return dev;
}

// group.cpp -- native multi-device driver behind the C-ABI (include/dinov2_hip.h, dinov2_hip_group_*).
//
// the path shards by independent images.  The caller's global batch [B, ...] is split contiguously (device g owns
// images [g*B/G, (g+1)*B/G), remainder to the low ranks) and every device writes its outputs straight into the caller's buffers at
// its shard offset.  No data-path collective.
//
// HOST BUFFERS WITHOUT A PCIe BUBBLE: each device runs `streams_per_device` (default 2) LANES -- a host thread + session
// (stream, workspace) + input staging buffer each.  A job (dinov2_hip_group_submit) goes to the lowest lane that has no job in
// flight -- so a caller that only ever has ONE job in flight (dinov2_hip_group_predict) stays on lane 0 and the other lanes never
// grow a workspace -- and every device works through three turnstiles in job order: host -> device copy of the shard, the forward (whole shard, one batch:
// splitting it would cost GEMM efficiency, measured), device -> host copy of the results.  With two jobs in flight lane B copies
// job k + 1 in while lane A computes job k, and lane A copies job k out while lane B computes job k + 1: the GPU only ever waits
// for the first copy-in and the last copy-out.  dinov2_hip_group_predict = submit + wait (one job in flight, nothing to overlap).
// Measured on one MI355X (tools/host_path.py, profiles/r03_host_path.json).
//
// This is what lets a C++ host of the reference's shape (/root/reference/inference.cpp:65, realtime.cpp:70 call dino_predict
// from one thread) use more than one GPU; bench.py's one-process-per-GPU torch.distributed run is the other way in.
//
// RCCL is loaded with dlopen on first use: single-GPU users of libdinov2_hip.so neither link nor load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sched.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "model.h"

namespace {

#define HIPG_TRY(expr)                                                                     \
    do {                                                                                   \
        const hipError_t e__ = (expr);                                                     \
        if (e__ != hipSuccess) {                                                           \
            set_err(err, errlen, "%s failed: %s", #expr, hipGetErrorString(e__));          \
            return DINOV2_HIP_ERR_HIP;                                                     \
        }                                                                                  \
    } while (0)

void set_err(char* err, size_t n, const char* fmt, ...) {
    if (!err || n == 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, n, fmt, ap);
    va_end(ap);
}

// ---- the five RCCL entry points this file needs, resolved at run time (rccl.h: ncclResult_t is an int enum, ncclSuccess = 0,
// ncclUint8 = 1, ncclComm_t an opaque pointer) ----
struct Rccl {
    void* so = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*Broadcast)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t st) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load(std::string* why) {
        if (so) return true;
        void* h = nullptr;
        std::string last;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
            const char* e = dlerror();  // ONE call: dlerror() clears the state it reports
            if (e) last = e;
        }
        if (!h) {
            *why = "cannot load librccl: " + (last.empty() ? std::string("?") : last);
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(h, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Broadcast || !GroupStart || !GroupEnd || !GetErrorString) {
            *why = "librccl lacks an expected symbol";
            CommInitAll = nullptr; CommDestroy = nullptr; Broadcast = nullptr;
            GroupStart = nullptr; GroupEnd = nullptr; GetErrorString = nullptr;
            dlclose(h);  // `so` stays null: the next call tries again instead of calling through null pointers
            return false;
        }
        so = h;
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// ---- host placement (round 6; unexercised on more than one physical GPU so far, hence defensive everywhere) ------------------------
// A device's PCIe attachment: bus id, NUMA node and the CPUs local to it, read from sysfs (absent or empty inside some containers: then
// nothing is known and nothing is bound).
struct DevPlace {
    std::string bdf, cpulist;
    int numa = -1;
};
DevPlace device_place(int device) {
    DevPlace p;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        return p;
    }
    for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
    p.bdf = bdf;
    auto slurp = [&](const char* leaf) {
        std::string out;
        const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/" + leaf;
        if (FILE* f = fopen(path.c_str(), "r")) {
            char buf[512];
            if (fgets(buf, sizeof buf, f)) out = buf;
            fclose(f);
        }
        while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
        return out;
    };
    const std::string nn = slurp("numa_node");
    if (!nn.empty()) p.numa = atoi(nn.c_str());
    p.cpulist = slurp("local_cpulist");
    return p;
}
// The worker threads of a device run on the CPUs local to it (they issue its copies and launches; a thread on the far socket pays a
// cross-socket hop per doorbell).  DINOV2_HIP_GROUP_NO_AFFINITY=1 leaves the threads where the scheduler puts them.
void bind_to_device_cpus(int device) {
    if (getenv("DINOV2_HIP_GROUP_NO_AFFINITY")) return;
    const DevPlace p = device_place(device);
    if (p.cpulist.empty()) return;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    const char* c = p.cpulist.c_str();
    int n = 0;
    while (*c) {  // "0-63,128-191"
        char* e = nullptr;
        const long a = strtol(c, &e, 10);
        if (e == c) break;
        long b = a;
        c = e;
        if (*c == '-') {
            b = strtol(c + 1, &e, 10);
            c = e;
        }
        for (long i = a; i <= b && i < CPU_SETSIZE; ++i)
            if (i >= 0 && CPU_ISSET((int)i, &allowed)) {
                CPU_SET((int)i, &want);
                ++n;
            }
        if (*c == ',') ++c;
    }
    if (n > 0) (void)sched_setaffinity(0, sizeof want, &want);  // (never an empty set: the container may expose none of the local CPUs)
}

struct Job {
    dinov2_hip_input in{};
    dinov2_hip_output out{};
    bool has_out = false;
    uint32_t flags = 0;
    int remaining = 0;  // ranks that have not finished it yet
    int lane = 0;       // the lane (of every rank) that runs it
    int rc = 0;
    char err[256] = {0};
};

}  // namespace

struct dinov2_hip_group {
    struct Lane {  // one host thread + session + input staging buffer
        dinov2_hip_session* session = nullptr;
        hipStream_t stream = nullptr;
        void* stage = nullptr;  // device copy of this lane's input images
        size_t stage_bytes = 0;
        std::deque<int64_t> q;  // tickets assigned to this lane, in submission order (guarded by the group's mu)
        std::thread th;
    };
    struct Rank {
        int device = 0;
        dinov2_hip_model* model = nullptr;
        std::vector<std::unique_ptr<Lane>> lanes;
        // the three turnstiles of a device, each passed in job order: copy-in, forward, copy-out
        std::mutex turn_mu;
        std::condition_variable turn_cv;
        int64_t turn[3] = {0, 0, 0};
    };
    std::vector<std::unique_ptr<Rank>> ranks;
    int nlanes = 1;
    double broadcast_ms = -1.0;  // < 0: every device read the file itself
    // job ring: job k lives in slot k % nlanes from submit until its wait returns (at most nlanes jobs are in flight); the lane that runs
    // it is chosen at submit: the lowest one without a job in flight (lane_busy)
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    int64_t submitted = 0;  // jobs handed in so far (tickets 0 .. submitted - 1)
    int64_t retired = 0;    // every ticket below this has been waited for (its slot is free)
    bool quit = false;
    std::vector<Job> ring;
    std::vector<int> lane_busy;  // jobs in flight (submitted, not waited for) per lane
    std::mutex call_mu;  // serialises dinov2_hip_group_predict callers (submit + wait as one unit)
};

namespace {

// images [lo, hi) of a global batch of B owned by rank r of G (same rule as dist.py: shard_range)
void shard_range(int B, int G, int r, int* lo, int* hi) {
    const int q = B / G, rem = B % G;
    *lo = r * q + (r < rem ? r : rem);
    *hi = *lo + q + (r < rem ? 1 : 0);
}

// Rank r's share of job `k` (a copy of the caller's descriptors), run by lane k % nlanes: the shard's images to the lane's
// staging buffer, the forward on device-resident input, the results to the caller's buffers at the shard offset -- each step behind
// the device's turnstile for it, so the steps of consecutive jobs overlap across lanes but never reorder.
void run_job(dinov2_hip_group* g, int r, int l, int64_t k, const Job& job, int* rc_out, char* err, size_t errlen) {
    auto& rk = *g->ranks[(size_t)r];
    auto& ln = *rk.lanes[(size_t)l];
    *rc_out = DINOV2_HIP_OK;
    auto enter = [&](int t) {
        std::unique_lock<std::mutex> lk(rk.turn_mu);
        rk.turn_cv.wait(lk, [&] { return rk.turn[t] == k; });
    };
    auto leave = [&](int t) {
        std::lock_guard<std::mutex> lk(rk.turn_mu);
        rk.turn[t] = k + 1;
        rk.turn_cv.notify_all();
    };
    const dinov2_hip_input& in = job.in;
    int lo, hi;
    shard_range(in.batch, (int)g->ranks.size(), r, &lo, &hi);
    if (hi <= lo) {  // B < G: nothing for this device, but the turnstiles still turn
        for (int t = 0; t < 3; ++t) {
            enter(t);
            leave(t);
        }
        return;
    }
    const dinov2_hip_hparams& hp = rk.model->hp;
    const bool classify = (job.flags & DINOV2_HIP_CLASSIFY) != 0;
    const bool raw = in.layout == DINOV2_HIP_U8_BGR_HWC;
    int32_t h = in.height, w = in.width;
    if (raw) dinov2_hip_preprocess_size(classify ? 1 : 0, in.height, in.width, (int32_t)hp.patch_size, &h, &w);
    const size_t in_stride = raw ? (size_t)in.height * in.width * 3 : (size_t)3 * h * w * sizeof(float);
    const size_t H = hp.hidden_size, C = hp.num_classes;
    const size_t P = (size_t)(h / (int)hp.patch_size) * (w / (int)hp.patch_size);
    const size_t tok_rows = P + (classify ? hp.num_register_tokens : 0);
    const size_t nbytes = (size_t)(hi - lo) * in_stride;
    const char* src = reinterpret_cast<const char*>(in.data) + (size_t)lo * in_stride;

    // ---- turnstile 0: host -> device (the lane's previous job has been fetched, so its staging buffer is free)
    enter(0);
    bool staged = true;
    if (nbytes > ln.stage_bytes) {
        if (ln.stage) (void)hipFree(ln.stage);
        ln.stage = nullptr;
        ln.stage_bytes = 0;
        if (hipMalloc(&ln.stage, nbytes) == hipSuccess) ln.stage_bytes = nbytes; else staged = false;
    }
    if (staged && (hipMemcpyAsync(ln.stage, src, nbytes, hipMemcpyHostToDevice, ln.stream) != hipSuccess ||
                   hipStreamSynchronize(ln.stream) != hipSuccess)) {
        (void)hipGetLastError();
        staged = false;  // (no staging memory: the session copies the images itself, inside turnstile 1)
    }
    leave(0);

    dinov2_hip_input si = in;
    si.data = reinterpret_cast<const float*>(staged ? (const char*)ln.stage : src);
    si.on_device = staged ? 1 : 0;
    si.batch = hi - lo;
    dinov2_hip_output so{};
    if (job.has_out) {
        so = job.out;
        if (so.cls) so.cls += (size_t)lo * H;
        if (so.patch_tokens) so.patch_tokens += (size_t)lo * tok_rows * H;
        if (so.logits) so.logits += (size_t)lo * C;
        if (so.probs) so.probs += (size_t)lo * C;
        if (so.topk_ids) so.topk_ids += (size_t)lo * (size_t)so.topk;
        if (so.topk_probs) so.topk_probs += (size_t)lo * (size_t)so.topk;
    }
    // ---- turnstile 1: the forward, results left in the session's workspace.  A shard longer than one pass takes (> 2^31-byte
    // activations) is computed in passes by dinov2_hip_predict and its results must leave pass by pass, inside this turnstile: decided
    // up front, so that the forward runs once either way.
    const bool in_passes = (size_t)si.batch > dinov2_max_pass_batch(rk.model, h, w);
    enter(1);
    int rc = dinov2_hip_predict(ln.session, &si, in_passes && job.has_out ? &so : nullptr, job.flags, err, errlen);
    if (rc == DINOV2_HIP_OK) rc = dinov2_hip_session_sync(ln.session);
    leave(1);
    // ---- turnstile 2: device -> host, under the next job's forward on the other lane
    enter(2);
    if (rc == DINOV2_HIP_OK && job.has_out && !in_passes) rc = dinov2_hip_fetch(ln.session, &so, err, errlen);
    leave(2);
    *rc_out = rc;
}

void worker(dinov2_hip_group* g, int r, int l) {
    (void)hipSetDevice(g->ranks[(size_t)r]->device);
    bind_to_device_cpus(g->ranks[(size_t)r]->device);
    auto& ln = *g->ranks[(size_t)r]->lanes[(size_t)l];
    for (;;) {
        Job job;
        int64_t k;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->quit || !ln.q.empty(); });
            if (ln.q.empty()) return;  // (quit: free waits for the jobs in flight first, so the queues are empty by then)
            k = ln.q.front();
            ln.q.pop_front();
            job = g->ring[(size_t)(k % g->nlanes)];
        }
        int rc = 0;
        char err[256] = {0};
        run_job(g, r, l, k, job, &rc, err, sizeof err);
        {
            std::lock_guard<std::mutex> lk(g->mu);
            Job& slot = g->ring[(size_t)(k % g->nlanes)];
            if (rc != DINOV2_HIP_OK && slot.rc == DINOV2_HIP_OK) {
                slot.rc = rc;
                snprintf(slot.err, sizeof slot.err, "device %d: %s", g->ranks[(size_t)r]->device, err[0] ? err : "predict failed");
            }
            if (--slot.remaining == 0) g->cv_done.notify_all();
        }
    }
}

}  // namespace

extern "C" void dinov2_hip_default_group_opts(dinov2_hip_group_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    dinov2_hip_default_load_opts(&o->load);
    o->n_devices = 0;
    o->devices = nullptr;
    o->broadcast = 1;
    o->streams_per_device = 2;
}

extern "C" void dinov2_hip_group_free(dinov2_hip_group* g) {
    if (!g) return;
    {
        // jobs still in flight (submitted, not waited for) run to completion first: a lane that left now would strand the other
        // lanes of its device at a turnstile
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv_done.wait(lk, [&] {
            for (int64_t k = g->retired; k < g->submitted; ++k)
                if (g->ring[(size_t)(k % g->nlanes)].remaining != 0) return false;
            return true;
        });
        g->quit = true;
    }
    g->cv_job.notify_all();
    for (auto& rk : g->ranks)
        for (auto& ln : rk->lanes)
            if (ln->th.joinable()) ln->th.join();
    for (auto& rk : g->ranks) {
        (void)hipSetDevice(rk->device);
        for (auto& ln : rk->lanes) {
            if (ln->session) dinov2_hip_session_free(ln->session);  // (synchronises the lane's stream first)
            if (ln->stage) (void)hipFree(ln->stage);
            if (ln->stream) (void)hipStreamDestroy(ln->stream);
        }
        if (rk->model) dinov2_hip_model_free(rk->model);
    }
    delete g;
}

extern "C" int dinov2_hip_group_create(const char* gguf_path, const dinov2_hip_group_opts* opts_in, dinov2_hip_group** out,
                                       char* err, size_t errlen) {
    if (!gguf_path || !out) {
        set_err(err, errlen, "null argument");
        return DINOV2_HIP_ERR_INVALID;
    }
    *out = nullptr;
    dinov2_hip_group_opts o;
    if (opts_in) o = *opts_in; else dinov2_hip_default_group_opts(&o);
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        set_err(err, errlen, "no HIP device visible");
        return DINOV2_HIP_ERR_HIP;
    }
    std::vector<int> devs;
    if (o.n_devices <= 0) {
        for (int d = 0; d < visible; ++d) devs.push_back(d);
    } else {
        if (o.n_devices > 64) {
            set_err(err, errlen, "n_devices %d is not plausible", o.n_devices);
            return DINOV2_HIP_ERR_INVALID;
        }
        for (int i = 0; i < o.n_devices; ++i) devs.push_back(o.devices ? o.devices[i] : i);
    }
    bool distinct = true;
    for (size_t i = 0; i < devs.size(); ++i) {
        if (devs[i] < 0 || devs[i] >= visible) {
            set_err(err, errlen, "device %d is not visible (%d devices)", devs[i], visible);
            return DINOV2_HIP_ERR_INVALID;
        }
        for (size_t j = 0; j < i; ++j) distinct = distinct && devs[j] != devs[i];
    }
    // RCCL wants one communicator rank per distinct device; a list that names a device twice (two sessions on one GPU: a
    // legitimate serving setup, and how the 1-GPU test box exercises the split) makes every rank read the file itself
    bool bcast = o.broadcast != 0 && distinct;
    if (bcast) {  // decided BEFORE ranks > 0 skip their tensor data: without librccl every rank reads the file itself
        std::string why;
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (!g_rccl.load(&why)) {
            if (getenv("DINOV2_HIP_GROUP_REQUIRE_RCCL")) {
                set_err(err, errlen, "%s", why.c_str());
                return DINOV2_HIP_ERR_HIP;
            }
            fprintf(stderr, "dinov2_hip_group_create: %s -- every device reads the GGUF itself\n", why.c_str());
            bcast = false;
        }
    }

    std::unique_ptr<dinov2_hip_group, void (*)(dinov2_hip_group*)> g(new dinov2_hip_group(), dinov2_hip_group_free);
    for (size_t i = 0; i < devs.size(); ++i) {
        std::unique_ptr<dinov2_hip_group::Rank> rk(new dinov2_hip_group::Rank());
        rk->device = devs[i];
        dinov2_hip_load_opts lo = o.load;
        lo.device = devs[i];
        lo.skip_tensor_data = (bcast && i > 0) ? 1 : 0;
        const int rc = dinov2_hip_model_load(gguf_path, &lo, &rk->model, err, errlen);
        if (rc != DINOV2_HIP_OK) return rc;
        g->ranks.push_back(std::move(rk));
    }
    if (bcast) {
        // Any failure in here -- no librccl, no communicator (peer access off, a fabric problem), a failed transfer -- degrades to "every
        // device reads the file itself" with a line on stderr, unless DINOV2_HIP_GROUP_REQUIRE_RCCL asks for an error: the weights are
        // identical either way, and this path has not yet run on more than one physical GPU.
        std::string fail;
        {
            std::lock_guard<std::mutex> lk(g_rccl_mu);  // communicator setup is process-global state in RCCL
            const int n = (int)devs.size();
            std::vector<void*> comms((size_t)n, nullptr);
            std::vector<hipStream_t> streams((size_t)n, nullptr);
            int nrc = 0;
            if (!g_rccl.load(&fail)) {
                nrc = -1;
            } else if ((nrc = g_rccl.CommInitAll(comms.data(), n, devs.data())) != 0) {
                fail = std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(nrc);
            }
            bool ok = nrc == 0;
            for (int i = 0; i < n && ok; ++i)
                ok = hipSetDevice(devs[(size_t)i]) == hipSuccess && hipStreamCreateWithFlags(&streams[(size_t)i], hipStreamNonBlocking) == hipSuccess;
            const auto t0 = std::chrono::steady_clock::now();
            if (ok) {
                // one message per rank: the whole arena (ViT-L f16 613 MB, ViT-g bf16 2.28 GB).  A ring broadcast over xGMI is bound
                // by one link (~153 GB/s), so few large messages, never many small ones.
                nrc = g_rccl.GroupStart();
                for (int i = 0; i < n && nrc == 0; ++i) {
                    (void)hipSetDevice(devs[(size_t)i]);
                    dinov2_hip_model* m = g->ranks[(size_t)i]->model;
                    nrc = g_rccl.Broadcast(m->arena, m->arena, m->arena_bytes, /*ncclUint8*/ 1, /*root*/ 0, comms[(size_t)i], streams[(size_t)i]);
                }
                const int erc = g_rccl.GroupEnd();
                if (nrc == 0) nrc = erc;
                for (int i = 0; i < n; ++i) {
                    (void)hipSetDevice(devs[(size_t)i]);
                    if (hipStreamSynchronize(streams[(size_t)i]) != hipSuccess) ok = false;
                }
                if (!ok || nrc != 0) fail = std::string("weight broadcast failed: ") + (nrc != 0 ? g_rccl.GetErrorString(nrc) : "HIP stream error");
            } else if (fail.empty()) {
                fail = "could not create the broadcast streams";
            }
            g->broadcast_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            for (int i = 0; i < n; ++i) {
                if (streams[(size_t)i]) {
                    (void)hipSetDevice(devs[(size_t)i]);
                    (void)hipStreamDestroy(streams[(size_t)i]);
                }
                if (comms[(size_t)i]) (void)g_rccl.CommDestroy(comms[(size_t)i]);
            }
            (void)hipGetLastError();
        }
        if (!fail.empty()) {
            if (getenv("DINOV2_HIP_GROUP_REQUIRE_RCCL")) {
                set_err(err, errlen, "%s", fail.c_str());
                return DINOV2_HIP_ERR_HIP;
            }
            fprintf(stderr, "dinov2_hip_group_create: %s -- every device reads the GGUF itself\n", fail.c_str());
            g->broadcast_ms = -1.0;
            for (size_t i = 1; i < devs.size(); ++i) {  // ranks > 0 were loaded without tensor data: load them again, with
                dinov2_hip_model_free(g->ranks[i]->model);
                g->ranks[i]->model = nullptr;
                dinov2_hip_load_opts lo = o.load;
                lo.device = devs[i];
                lo.skip_tensor_data = 0;
                const int rc = dinov2_hip_model_load(gguf_path, &lo, &g->ranks[i]->model, err, errlen);
                if (rc != DINOV2_HIP_OK) return rc;
            }
        }
    }
    g->nlanes = o.streams_per_device <= 0 ? 2 : o.streams_per_device > 4 ? 4 : o.streams_per_device;
    g->ring.resize((size_t)g->nlanes);
    g->lane_busy.assign((size_t)g->nlanes, 0);
    for (auto& rk : g->ranks) {
        HIPG_TRY(hipSetDevice(rk->device));
        int least = 0, greatest = 0;  // numerically lower = higher priority
        HIPG_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        for (int l = 0; l < g->nlanes; ++l) {
            std::unique_ptr<dinov2_hip_group::Lane> ln(new dinov2_hip_group::Lane());
            (void)least;
            HIPG_TRY(hipStreamCreateWithPriority(&ln->stream, hipStreamNonBlocking, greatest));
            rk->lanes.push_back(std::move(ln));
            auto& lane = *rk->lanes.back();
            const int rc = dinov2_hip_session_create(rk->model, (void*)lane.stream, &lane.session, err, errlen);
            if (rc != DINOV2_HIP_OK) return rc;
        }
    }
    for (size_t i = 0; i < g->ranks.size(); ++i)
        for (int l = 0; l < g->nlanes; ++l) g->ranks[i]->lanes[(size_t)l]->th = std::thread(worker, g.get(), (int)i, l);
    *out = g.release();
    return DINOV2_HIP_OK;
}

extern "C" int dinov2_hip_group_size(const dinov2_hip_group* g) { return g ? (int)g->ranks.size() : 0; }

extern "C" dinov2_hip_model* dinov2_hip_group_model(dinov2_hip_group* g, int32_t rank) {
    return g && rank >= 0 && (size_t)rank < g->ranks.size() ? g->ranks[(size_t)rank]->model : nullptr;
}

extern "C" double dinov2_hip_group_broadcast_ms(const dinov2_hip_group* g) { return g ? g->broadcast_ms : -1.0; }

// One line per device: ordinal, PCI bus id, NUMA node, local CPUs (what the device's worker threads are bound to), which of the group's
// other devices it can reach peer to peer, and how the weights got there.  For logs and for callers that place their page-locked
// buffers (dinov2_hip_host_alloc from a thread bound to the device's CPUs lands on its node by first touch).
extern "C" int dinov2_hip_group_describe(const dinov2_hip_group* g, char* out, size_t cap) {
    if (!g || !out || cap == 0) return DINOV2_HIP_ERR_INVALID;
    std::string s;
    for (size_t i = 0; i < g->ranks.size(); ++i) {
        const int d = g->ranks[i]->device;
        const DevPlace p = device_place(d);
        char line[512];
        snprintf(line, sizeof line, "device %d pci %s numa %d cpus %s peers", d, p.bdf.empty() ? "?" : p.bdf.c_str(), p.numa,
                 p.cpulist.empty() ? "?" : p.cpulist.c_str());
        s += line;
        for (size_t j = 0; j < g->ranks.size(); ++j) {
            const int e = g->ranks[j]->device;
            int can = 0;
            if (e != d && hipDeviceCanAccessPeer(&can, d, e) != hipSuccess) {
                (void)hipGetLastError();
                can = -1;
            }
            s += e == d ? " ." : can > 0 ? " y" : can == 0 ? " n" : " ?";
        }
        s += g->broadcast_ms >= 0 ? (i == 0 ? " weights file" : " weights rccl-broadcast") : " weights file";
        s += "\n";
    }
    snprintf(out, cap, "%s", s.c_str());
    return DINOV2_HIP_OK;
}

// `require_empty`: refuse (under the SAME hold of g->mu that would enqueue) when another ticket is still un-waited -- the blocking
// dinov2_hip_group_predict's guard; checking it in one critical section and enqueuing in another let a concurrent submit slip in between
// and orphan predict's ticket
static int group_submit_impl(dinov2_hip_group* g, const dinov2_hip_input* in, const dinov2_hip_output* out, uint32_t flags, int64_t* ticket,
                             bool require_empty, char* err, size_t errlen) {
    if (!g || !in || !in->data || in->batch <= 0 || !ticket) {
        set_err(err, errlen, "null group / input / ticket");
        return DINOV2_HIP_ERR_INVALID;
    }
    if (in->on_device || (out && out->on_device)) {
        // a device pointer belongs to ONE device; the group's contract is the reference's: host images in, host results out
        set_err(err, errlen, "group predict takes host buffers (each device copies its own shard)");
        return DINOV2_HIP_ERR_INVALID;
    }
    {
        // layout / size checks BEFORE anything is queued: turnstile 0 sizes its host -> device copy from these fields, ahead of
        // dinov2_hip_predict's own checks
        const int vrc = dinov2_check_input(g->ranks[0]->model, in, err, errlen);
        if (vrc != DINOV2_HIP_OK) return vrc;
        if ((flags & DINOV2_HIP_CLASSIFY) && !g->ranks[0]->model->hp.has_classifier) {
            set_err(err, errlen, "classify requested but the model was loaded without a classifier head");
            return DINOV2_HIP_ERR_NO_HEAD;
        }
        if (out && out->on_device == 0 && out->topk < 0) {
            set_err(err, errlen, "negative topk");
            return DINOV2_HIP_ERR_INVALID;
        }
    }
    std::unique_lock<std::mutex> lk(g->mu);
    if (require_empty && g->submitted != g->retired) {
        // submit + wait as one unit needs an empty pipeline: with an un-waited ticket ahead of it, the wait would be refused ("in submission
        // order") AFTER the job had been queued -- a ticket nobody holds, writing into the caller's buffers
        set_err(err, errlen, "%lld submitted job(s) not waited for yet: dinov2_hip_group_wait them before dinov2_hip_group_predict",
                (long long)(g->submitted - g->retired));
        return DINOV2_HIP_ERR_INVALID;
    }
    if (g->submitted - g->retired >= g->nlanes) {
        set_err(err, errlen, "%d jobs already in flight (streams_per_device): wait for one first", g->nlanes);
        return DINOV2_HIP_ERR_INVALID;
    }
    const int64_t k = g->submitted;
    int lane = 0;
    while (lane < g->nlanes - 1 && g->lane_busy[(size_t)lane] != 0) ++lane;  // (fewer than nlanes in flight: a free lane exists)
    Job& slot = g->ring[(size_t)(k % g->nlanes)];
    slot = Job{};
    slot.lane = lane;
    ++g->lane_busy[(size_t)lane];
    for (auto& rk : g->ranks) rk->lanes[(size_t)lane]->q.push_back(k);
    slot.in = *in;
    if (out) slot.out = *out;
    slot.has_out = out != nullptr;
    slot.flags = flags;
    slot.remaining = (int)g->ranks.size();
    ++g->submitted;
    *ticket = k;
    lk.unlock();
    g->cv_job.notify_all();
    return DINOV2_HIP_OK;
}

extern "C" int dinov2_hip_group_submit(dinov2_hip_group* g, const dinov2_hip_input* in, const dinov2_hip_output* out, uint32_t flags,
                                       int64_t* ticket, char* err, size_t errlen) {
    return group_submit_impl(g, in, out, flags, ticket, false, err, errlen);
}

extern "C" int dinov2_hip_group_wait(dinov2_hip_group* g, int64_t ticket, char* err, size_t errlen) {
    if (!g) return DINOV2_HIP_ERR_INVALID;
    std::unique_lock<std::mutex> lk(g->mu);
    if (ticket != g->retired || ticket >= g->submitted) {  // in submission order: results land in the order the images came in
        set_err(err, errlen, "wait for tickets in submission order (next: %lld)", (long long)g->retired);
        return DINOV2_HIP_ERR_INVALID;
    }
    Job& slot = g->ring[(size_t)(ticket % g->nlanes)];
    g->cv_done.wait(lk, [&] { return slot.remaining == 0; });
    const int rc = slot.rc;
    if (rc != DINOV2_HIP_OK) set_err(err, errlen, "%s", slot.err);
    --g->lane_busy[(size_t)slot.lane];
    ++g->retired;
    return rc;
}

extern "C" int dinov2_hip_group_predict(dinov2_hip_group* g, const dinov2_hip_input* in, dinov2_hip_output* out, uint32_t flags,
                                        char* err, size_t errlen) {
    if (!g) {
        set_err(err, errlen, "null group");
        return DINOV2_HIP_ERR_INVALID;
    }
    std::lock_guard<std::mutex> call(g->call_mu);
    int64_t t = 0;
    const int rc = group_submit_impl(g, in, out, flags, &t, true, err, errlen);  // (emptiness check and enqueue under one hold of g->mu)
    if (rc != DINOV2_HIP_OK) return rc;
    return dinov2_hip_group_wait(g, t, err, errlen);
}

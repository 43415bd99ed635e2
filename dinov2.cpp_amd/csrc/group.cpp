// group.cpp -- native multi-device driver behind the C-ABI (include/dinov2_hip.h, dinov2_hip_group_*).
//
// the path shards by independent images.  One HOST THREAD + one session (stream + workspace) per device, the
// caller's global batch [B, ...] split contiguously (device g owns images [g*B/G, (g+1)*B/G), remainder to the low ranks) and
// every device writing its outputs straight into the caller's buffers at its shard offset.  No data-path collective.  The ONE
// collective is the load-time broadcast of rank 0's converted weight arena over xGMI: single-process RCCL
// (ncclCommInitAll + ncclBroadcast inside a group call), so the GGUF is parsed / dequantised once instead of G times.
//
// This is what lets a C++ host of the reference's shape (/root/reference/inference.cpp:65, realtime.cpp:70 call dino_predict
// from one thread) use more than one GPU; bench.py's one-process-per-GPU torch.distributed run is the other way in.
//
// RCCL is loaded with dlopen on first use: single-GPU users of libdinov2_hip.so neither link nor load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "model.h"

namespace {

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

struct Job {
    const dinov2_hip_input* in = nullptr;
    const dinov2_hip_output* out = nullptr;
    uint32_t flags = 0;
};

}  // namespace

struct dinov2_hip_group {
    struct Rank {
        int device = 0;
        dinov2_hip_model* model = nullptr;
        dinov2_hip_session* session = nullptr;
        std::thread th;
        int rc = 0;
        char err[256] = {0};
    };
    std::vector<std::unique_ptr<Rank>> ranks;
    double broadcast_ms = -1.0;  // < 0: every device read the file itself
    // one job at a time, handed to all workers; generation counters instead of per-call thread creation
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    Job job;
    std::mutex call_mu;  // dinov2_hip_group_predict is not re-entrant on one group
};

namespace {

// images [lo, hi) of a global batch of B owned by rank r of G (same rule as dist.py: shard_range)
void shard_range(int B, int G, int r, int* lo, int* hi) {
    const int q = B / G, rem = B % G;
    *lo = r * q + (r < rem ? r : rem);
    *hi = *lo + q + (r < rem ? 1 : 0);
}

// the worker's share of one group predict: pointers advanced to the shard offset, then the ordinary single-device entry point
void run_shard(dinov2_hip_group* g, int r, const Job& job) {
    auto& rk = *g->ranks[(size_t)r];
    rk.rc = DINOV2_HIP_OK;
    rk.err[0] = 0;
    const dinov2_hip_input& in = *job.in;
    int lo, hi;
    shard_range(in.batch, (int)g->ranks.size(), r, &lo, &hi);
    if (hi <= lo) return;
    const dinov2_hip_hparams& hp = rk.model->hp;
    const bool classify = (job.flags & DINOV2_HIP_CLASSIFY) != 0;
    const bool raw = in.layout == DINOV2_HIP_U8_BGR_HWC;
    int32_t h = in.height, w = in.width;
    if (raw) dinov2_hip_preprocess_size(classify ? 1 : 0, in.height, in.width, (int32_t)hp.patch_size, &h, &w);
    const size_t in_stride = raw ? (size_t)in.height * in.width * 3 : (size_t)3 * h * w * sizeof(float);
    const size_t H = hp.hidden_size, C = hp.num_classes;
    const size_t P = (size_t)(h / (int)hp.patch_size) * (w / (int)hp.patch_size);
    const size_t tok_rows = P + (classify ? hp.num_register_tokens : 0);
    dinov2_hip_input si = in;
    si.data = reinterpret_cast<const float*>(reinterpret_cast<const char*>(in.data) + (size_t)lo * in_stride);
    si.batch = hi - lo;
    dinov2_hip_output so{};
    if (job.out) {
        so = *job.out;
        if (so.cls) so.cls += (size_t)lo * H;
        if (so.patch_tokens) so.patch_tokens += (size_t)lo * tok_rows * H;
        if (so.logits) so.logits += (size_t)lo * C;
        if (so.probs) so.probs += (size_t)lo * C;
        if (so.topk_ids) so.topk_ids += (size_t)lo * (size_t)so.topk;
        if (so.topk_probs) so.topk_probs += (size_t)lo * (size_t)so.topk;
    }
    rk.rc = dinov2_hip_predict(rk.session, &si, job.out ? &so : nullptr, job.flags, rk.err, sizeof rk.err);
    if (rk.rc == DINOV2_HIP_OK && (!job.out || job.out->on_device)) rk.rc = dinov2_hip_session_sync(rk.session);
}

void worker(dinov2_hip_group* g, int r) {
    (void)hipSetDevice(g->ranks[(size_t)r]->device);
    uint64_t seen = 0;
    for (;;) {
        Job job;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->quit || g->gen != seen; });
            if (g->quit) return;
            seen = g->gen;
            job = g->job;
        }
        run_shard(g, r, job);
        {
            std::lock_guard<std::mutex> lk(g->mu);
            if (--g->pending == 0) g->cv_done.notify_all();
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
}

extern "C" void dinov2_hip_group_free(dinov2_hip_group* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
    }
    g->cv_job.notify_all();
    for (auto& rk : g->ranks)
        if (rk->th.joinable()) rk->th.join();
    for (auto& rk : g->ranks) {
        if (rk->session) dinov2_hip_session_free(rk->session);
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
        std::string why;
        std::lock_guard<std::mutex> lk(g_rccl_mu);  // communicator setup is process-global state in RCCL
        if (!g_rccl.load(&why)) {
            set_err(err, errlen, "%s", why.c_str());
            return DINOV2_HIP_ERR_HIP;
        }
        const int n = (int)devs.size();
        std::vector<void*> comms((size_t)n, nullptr);
        std::vector<hipStream_t> streams((size_t)n, nullptr);
        int nrc = g_rccl.CommInitAll(comms.data(), n, devs.data());
        if (nrc != 0) {
            set_err(err, errlen, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(nrc));
            return DINOV2_HIP_ERR_HIP;
        }
        bool ok = true;
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
        }
        g->broadcast_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < n; ++i) {
            if (streams[(size_t)i]) {
                (void)hipSetDevice(devs[(size_t)i]);
                (void)hipStreamDestroy(streams[(size_t)i]);
            }
            if (comms[(size_t)i]) (void)g_rccl.CommDestroy(comms[(size_t)i]);
        }
        if (!ok || nrc != 0) {
            set_err(err, errlen, "weight broadcast failed: %s", nrc != 0 ? g_rccl.GetErrorString(nrc) : "HIP stream error");
            return DINOV2_HIP_ERR_HIP;
        }
    }
    for (auto& rk : g->ranks) {
        const int rc = dinov2_hip_session_create(rk->model, nullptr, &rk->session, err, errlen);
        if (rc != DINOV2_HIP_OK) return rc;
    }
    for (size_t i = 0; i < g->ranks.size(); ++i) g->ranks[i]->th = std::thread(worker, g.get(), (int)i);
    *out = g.release();
    return DINOV2_HIP_OK;
}

extern "C" int dinov2_hip_group_size(const dinov2_hip_group* g) { return g ? (int)g->ranks.size() : 0; }

extern "C" dinov2_hip_model* dinov2_hip_group_model(dinov2_hip_group* g, int32_t rank) {
    return g && rank >= 0 && (size_t)rank < g->ranks.size() ? g->ranks[(size_t)rank]->model : nullptr;
}

extern "C" double dinov2_hip_group_broadcast_ms(const dinov2_hip_group* g) { return g ? g->broadcast_ms : -1.0; }

extern "C" int dinov2_hip_group_predict(dinov2_hip_group* g, const dinov2_hip_input* in, dinov2_hip_output* out, uint32_t flags,
                                        char* err, size_t errlen) {
    if (!g || !in || !in->data || in->batch <= 0) {
        set_err(err, errlen, "null group / input");
        return DINOV2_HIP_ERR_INVALID;
    }
    if (in->on_device || (out && out->on_device)) {
        // a device pointer belongs to ONE device; the group's contract is the reference's: host images in, host results out
        set_err(err, errlen, "group predict takes host buffers (each device copies its own shard)");
        return DINOV2_HIP_ERR_INVALID;
    }
    std::lock_guard<std::mutex> call(g->call_mu);
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->job = Job{in, out, flags};
        g->pending = (int)g->ranks.size();
        ++g->gen;
    }
    g->cv_job.notify_all();
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv_done.wait(lk, [&] { return g->pending == 0; });
    }
    for (auto& rk : g->ranks)
        if (rk->rc != DINOV2_HIP_OK) {
            set_err(err, errlen, "device %d: %s", rk->device, rk->err[0] ? rk->err : "predict failed");
            return rk->rc;
        }
    return DINOV2_HIP_OK;
}

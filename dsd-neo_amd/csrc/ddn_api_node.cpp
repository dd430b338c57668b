// ddn_api_node.cpp - all the GPUs of one node from C (include/ddn_node.h): block partition of the channel index, one chain object
// and one host thread per device.  Host-only code; no collective on the data path.
// (round 6) generic over the chain objects of include/ddn_chain.h: a part's chain is reached through a small table of operations
// (create / run from host memory / run from device memory / wait / flush / destroy), so the same driver runs the P25 Phase 1 chain,
// the mixed P25 + DMR + NXDN48 object of BASELINE configs[3] (ddn_mixed_partition decides each device's three groups), one DMR /
// NXDN / M17 / YSF chain, or the P25 Phase 2 chain.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "ddn_hip.h"
#include "ddn_internal.h"
#include "ddn_node.h"

namespace {
enum Cmd { C_NONE = 0, C_CREATE, C_RUN_HOST, C_RUN_DEV, C_WAIT, C_FLUSH, C_ALLOC, C_UPLOAD, C_DOWNLOAD, C_FREE, C_CALL, C_QUIT };

struct Part;
struct Ops { // what a node asks of a part's chain object, whatever its kind (called on the part's thread, its device current)
    int (*create)(Part*);
    int (*run_host)(Part*);   // a_iq3: host pointers of this part's blocks, a_out: result buffers (P25 Phase 1 only)
    int (*run_dev)(Part*);    // a_iq3: device pointers
    int (*wait)(Part*);
    int (*flush)(Part*);
    void (*destroy)(Part*);
};

struct Part {
    int device = 0, first = 0, count = 0;
    int32_t first3[3] = {0, 0, 0}, count3[3] = {0, 0, 0}; // mixed: this part's block of each group
    int set_device_rc = DDN_OK;
    const Ops* ops = nullptr;
    void* chain = nullptr; // ddn_p25_chain* / ddn_mixed_chain* / ddn_fsk4_chain* / ddn_p25p2_chain*
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::mutex call_mu; // one caller at a time per part: the argument fields below belong to the call in flight
    Cmd cmd = C_NONE;
    bool done = true;
    int rc = DDN_OK;
    // arguments of the pending command
    const void* a_iq3[3] = {nullptr, nullptr, nullptr};
    const ddn_p25_chain_host_out* a_out = nullptr;
    size_t a_bytes = 0;
    void* a_dst = nullptr;
    const void* a_src = nullptr;
    void** a_pp = nullptr;
    int (*a_fn)(void*, void*) = nullptr;
    void* a_arg = nullptr;
    // configuration of this part's chain, by kind
    ddn_p25_chain_config ccfg;
    ddn_mixed_chain_config mcfg;
    ddn_fsk4_chain_config fcfg;
    ddn_p25p2_chain_config p2cfg;
    const uint64_t* p2seed = nullptr;
    // the kinds without a host-memory call of their own: device input buffers of this part (two, used in turn) and the stream the
    // copies and the chain run on
    void* d_in[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    size_t in_bytes[3] = {0, 0, 0};
    int in_turn = 0;
    int in_flight = 0; // mixed, host input: calls queued since the last wait (a buffer set is reused every second call)
    hipStream_t st = nullptr;
    size_t sample_bytes = 2;
    int samples = 0;
    char err[256] = {0};
};

int
hip_rc(hipError_t e, const char* what) {
    if (e == hipSuccess) {
        return DDN_OK;
    }
    ddn_set_error("%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? DDN_ENOMEM : DDN_EHIP;
}

// host blocks -> this part's device input buffers (two sets, used in turn: the previous call's kernels may still read the other set),
// on the part's stream ahead of the run call
int
stage_inputs(Part* p, int n_groups, const void* d_out[3]) {
    for (int g = 0; g < n_groups; g++) {
        d_out[g] = nullptr;
        if (p->in_bytes[g] == 0) {
            continue;
        }
        if (!p->a_iq3[g]) {
            ddn_set_error("ddn_node_run_host: no input for a group that has channels");
            return DDN_EINVAL;
        }
        void*& d = p->d_in[g][p->in_turn];
        if (!d) {
            const int rc = hip_rc(hipMalloc(&d, p->in_bytes[g]), "hipMalloc (node input)");
            if (rc != DDN_OK) {
                return rc;
            }
        }
        const int rc = hip_rc(hipMemcpyAsync(d, p->a_iq3[g], p->in_bytes[g], hipMemcpyHostToDevice, p->st), "hipMemcpyAsync H2D");
        if (rc != DDN_OK) {
            return rc;
        }
        d_out[g] = d;
    }
    p->in_turn ^= 1;
    return DDN_OK;
}

int
make_stream(Part* p) {
    return p->st ? DDN_OK : hip_rc(hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking), "hipStreamCreate");
}

void
free_inputs(Part* p) {
    for (auto& g : p->d_in) {
        for (void*& d : g) {
            (void)hipFree(d);
            d = nullptr;
        }
    }
    if (p->st) {
        (void)hipStreamDestroy(p->st);
        p->st = nullptr;
    }
}

// ---- P25 Phase 1 --------------------------------------------------------------------------------------------------------------
const Ops kP25 = {
    [](Part* p) {
        ddn_p25_chain* c = nullptr;
        int rc = ddn_p25_chain_create(&p->ccfg, &c);
        if (rc == DDN_OK) {
            rc = ddn_p25_chain_set_first_channel(c, p->first);
        }
        p->chain = c;
        return rc;
    },
    [](Part* p) { return ddn_p25_chain_run_host((ddn_p25_chain*)p->chain, p->a_iq3[0], p->a_out); },
    [](Part* p) { return ddn_p25_chain_run_pipelined((ddn_p25_chain*)p->chain, p->a_iq3[0]); },
    [](Part* p) { return ddn_p25_chain_wait((ddn_p25_chain*)p->chain); },
    [](Part* p) { return ddn_p25_chain_flush((ddn_p25_chain*)p->chain); },
    [](Part* p) { ddn_p25_chain_destroy((ddn_p25_chain*)p->chain); },
};

// ---- mixed: P25 Phase 1 + DMR + NXDN48 groups ------------------------------------------------------------------------------------
int
mixed_flush(Part* p) {
    ddn_mixed_chain* m = (ddn_mixed_chain*)p->chain;
    p->in_flight = 0;
    int rc = ddn_mixed_chain_wait(m);
    if (rc == DDN_OK && p->count3[0] > 0) {
        rc = ddn_p25_chain_flush((ddn_p25_chain*)ddn_mixed_chain_part(m, 0));
    }
    for (int g = 1; g < 3 && rc == DDN_OK; g++) {
        if (p->count3[g] > 0) {
            rc = ddn_fsk4_chain_flush((ddn_fsk4_chain*)ddn_mixed_chain_part(m, g), nullptr);
        }
    }
    if (rc == DDN_OK) {
        rc = hip_rc(hipDeviceSynchronize(), "hipDeviceSynchronize");
    }
    return rc;
}
const Ops kMixed = {
    [](Part* p) {
        ddn_mixed_chain* m = nullptr;
        int rc = ddn_mixed_chain_create(&p->mcfg, &m);
        p->chain = m;
        if (rc == DDN_OK && p->count3[0] > 0) {
            rc = ddn_p25_chain_set_first_channel((ddn_p25_chain*)ddn_mixed_chain_part(m, 0), p->first3[0]);
        }
        return rc;
    },
    [](Part* p) {
        int rc = make_stream(p);
        const void* d[3];
        if (rc == DDN_OK && p->in_flight >= 2) {
            // the mixed object frees a call's input only at its _wait, and this call's copy goes into the buffer set of the call
            // before the last: wait every second call (the other set's call keeps the device busy meanwhile)
            rc = ddn_mixed_chain_wait((ddn_mixed_chain*)p->chain);
            p->in_flight = 0;
        }
        if (rc == DDN_OK) {
            rc = stage_inputs(p, 3, d);
        }
        p->in_flight++;
        if (rc == DDN_OK) { // (the mixed object runs on streams of its own: the copies have to be in before it is queued)
            rc = hip_rc(hipStreamSynchronize(p->st), "hipStreamSynchronize");
        }
        return rc == DDN_OK ? ddn_mixed_chain_run((ddn_mixed_chain*)p->chain, d[0], d[1], d[2]) : rc;
    },
    [](Part* p) { return ddn_mixed_chain_run((ddn_mixed_chain*)p->chain, p->a_iq3[0], p->a_iq3[1], p->a_iq3[2]); },
    [](Part* p) {
        p->in_flight = 0;
        return ddn_mixed_chain_wait((ddn_mixed_chain*)p->chain);
    },
    mixed_flush,
    [](Part* p) { ddn_mixed_chain_destroy((ddn_mixed_chain*)p->chain); },
};

// ---- one DMR / NXDN / M17 / YSF chain ------------------------------------------------------------------------------------------
const Ops kFsk4 = {
    [](Part* p) {
        ddn_fsk4_chain* c = nullptr;
        int rc = ddn_fsk4_chain_create(&p->fcfg, &c);
        p->chain = c;
        return rc == DDN_OK ? make_stream(p) : rc;
    },
    [](Part* p) {
        const void* d[3];
        const int rc = stage_inputs(p, 1, d);
        return rc == DDN_OK ? ddn_fsk4_chain_run((ddn_fsk4_chain*)p->chain, d[0], p->st) : rc;
    },
    [](Part* p) { return ddn_fsk4_chain_run((ddn_fsk4_chain*)p->chain, p->a_iq3[0], p->st); },
    [](Part* p) { return hip_rc(hipStreamSynchronize(p->st), "hipStreamSynchronize"); },
    [](Part* p) {
        const int rc = ddn_fsk4_chain_flush((ddn_fsk4_chain*)p->chain, p->st);
        return rc == DDN_OK ? hip_rc(hipStreamSynchronize(p->st), "hipStreamSynchronize") : rc;
    },
    [](Part* p) { ddn_fsk4_chain_destroy((ddn_fsk4_chain*)p->chain); },
};

// ---- P25 Phase 2 ----------------------------------------------------------------------------------------------------------------
const Ops kP25p2 = {
    [](Part* p) {
        ddn_p25p2_chain* c = nullptr;
        int rc = ddn_p25p2_chain_create(&p->p2cfg, p->p2seed ? p->p2seed + p->first : nullptr, &c);
        p->chain = c;
        return rc == DDN_OK ? make_stream(p) : rc;
    },
    [](Part* p) {
        const void* d[3];
        const int rc = stage_inputs(p, 1, d);
        return rc == DDN_OK ? ddn_p25p2_chain_run((ddn_p25p2_chain*)p->chain, d[0], p->st) : rc;
    },
    [](Part* p) { return ddn_p25p2_chain_run((ddn_p25p2_chain*)p->chain, p->a_iq3[0], p->st); },
    [](Part* p) { return hip_rc(hipStreamSynchronize(p->st), "hipStreamSynchronize"); },
    [](Part* p) {
        const int rc = ddn_p25p2_chain_flush((ddn_p25p2_chain*)p->chain, p->st);
        return rc == DDN_OK ? hip_rc(hipStreamSynchronize(p->st), "hipStreamSynchronize") : rc;
    },
    [](Part* p) { ddn_p25p2_chain_destroy((ddn_p25p2_chain*)p->chain); },
};
} // namespace

struct ddn_node {
    ddn_node_config cfg;
    std::vector<Part*> parts;
    size_t sample_bytes;
    ddn_fsk4_chain_config fsk4_copy;   // (the caller's structures need not outlive ddn_node_create)
    ddn_p25p2_chain_config p25p2_copy;
};

static void
part_main(Part* p) {
    // a part that cannot make its device current must not run on whatever device is current instead (it would share device 0
    // with that device's own part while ddn_node_part_info reports another): every command then fails with DDN_ENODEV
    if (hipSetDevice(p->device) != hipSuccess) {
        p->set_device_rc = DDN_ENODEV;
    }
    for (;;) {
        Cmd c;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return !p->done; });
            c = p->cmd;
        }
        int rc = DDN_OK;
        if (p->set_device_rc != DDN_OK && c != C_QUIT) {
            ddn_set_error("ddn_node: hipSetDevice(%d) failed", p->device);
            rc = p->set_device_rc;
        } else if ((c == C_RUN_HOST || c == C_RUN_DEV || c == C_WAIT || c == C_FLUSH || c == C_CALL) && !p->chain) {
            ddn_set_error("ddn_node: part without a chain object");
            rc = DDN_EINVAL;
        } else {
            switch (c) {
                case C_CREATE: rc = p->ops->create(p); break;
                case C_RUN_HOST: rc = p->ops->run_host(p); break;
                case C_RUN_DEV: rc = p->ops->run_dev(p); break;
                case C_WAIT: rc = p->ops->wait(p); break;
                case C_FLUSH: rc = p->ops->flush(p); break;
                case C_CALL: rc = p->a_fn(p->chain, p->a_arg); break;
                case C_ALLOC: rc = hip_rc(hipMalloc(p->a_pp, p->a_bytes), "hipMalloc"); break;
                case C_UPLOAD: rc = hip_rc(hipMemcpy(p->a_dst, p->a_src, p->a_bytes, hipMemcpyHostToDevice), "hipMemcpy H2D"); break;
                case C_DOWNLOAD: rc = hip_rc(hipMemcpy(p->a_dst, p->a_src, p->a_bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H"); break;
                case C_FREE: (void)hipFree(p->a_dst); break;
                case C_QUIT:
                    if (p->chain) {
                        p->ops->destroy(p);
                        p->chain = nullptr;
                    }
                    free_inputs(p);
                    break;
                default: break;
            }
        }
        if (rc != DDN_OK) { // the error text is per thread: keep this thread's
            strncpy(p->err, ddn_last_error(), sizeof(p->err) - 1);
        }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->rc = rc;
            p->done = true;
        }
        p->cv.notify_all();
        if (c == C_QUIT) {
            return;
        }
    }
}

static void
post(Part* p, Cmd c) {
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->cmd = c;
        p->done = false;
    }
    p->cv.notify_all();
}

static int
join(Part* p) {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return p->done; });
    return p->rc;
}

// one command on every part (the callers have filled the parts' argument fields under the parts' call locks, taken in part order)
static int
all_locked(ddn_node* n, Cmd c) {
    for (Part* p : n->parts) {
        post(p, c);
    }
    int rc = DDN_OK;
    for (Part* p : n->parts) {
        const int r = join(p);
        if (r != DDN_OK && rc == DDN_OK) {
            rc = r;
            ddn_set_error("ddn_node: device %d (channels %d..%d): %s", p->device, p->first, p->first + p->count - 1, p->err);
        }
    }
    return rc;
}

namespace {
struct LockAll { // every part's call lock, in part order (two host threads calling into one node take turns)
    ddn_node* n;
    explicit LockAll(ddn_node* node) : n(node) {
        for (Part* p : n->parts) {
            p->call_mu.lock();
        }
    }
    ~LockAll() {
        for (auto it = n->parts.rbegin(); it != n->parts.rend(); ++it) {
            (*it)->call_mu.unlock();
        }
    }
    LockAll(const LockAll&) = delete;
    LockAll& operator=(const LockAll&) = delete;
};
} // namespace

extern "C" int
ddn_node_partition(int n_channels, int rank, int world, int* first, int* count) {
    if (n_channels < 0 || world <= 0 || rank < 0 || rank >= world || !first || !count) {
        return DDN_EINVAL;
    }
    const int base = n_channels / world, extra = n_channels % world;
    *first = rank * base + (rank < extra ? rank : extra);
    *count = base + (rank < extra ? 1 : 0);
    return DDN_OK;
}

static int
all(ddn_node* n, Cmd c) {
    LockAll lk(n);
    return all_locked(n, c);
}

extern "C" void
ddn_node_destroy(ddn_node* n) {
    if (!n) {
        return;
    }
    for (Part* p : n->parts) {
        if (p->th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(p->call_mu);
                post(p, C_QUIT);
                (void)join(p);
            }
            p->th.join();
        }
        delete p;
    }
    delete n;
}

extern "C" int
ddn_node_create(const ddn_node_config* cfg, ddn_node** out) {
    if (!cfg || !out || cfg->samples_per_call <= 0 || cfg->block_len <= 0 || cfg->n_devices < 0 || cfg->n_channels < 0) {
        ddn_set_error("ddn_node_create: bad configuration");
        return DDN_EINVAL;
    }
    const int kind = cfg->kind;
    const int total = kind == DDN_NODE_MIXED ? cfg->n_channels + cfg->n_dmr + cfg->n_nxdn48 : cfg->n_channels;
    if (kind < DDN_NODE_P25 || kind > DDN_NODE_P25P2 || total <= 0 || (kind == DDN_NODE_MIXED && (cfg->n_dmr < 0 || cfg->n_nxdn48 < 0))
        || (kind == DDN_NODE_FSK4 && !cfg->fsk4) || (kind == DDN_NODE_P25P2 && !cfg->p25p2)) {
        ddn_set_error("ddn_node_create: bad configuration (kind %d)", kind);
        return DDN_EINVAL;
    }
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        ddn_set_error("ddn_node_create: no HIP device");
        return DDN_ENODEV;
    }
    int parts = cfg->n_devices > 0 ? cfg->n_devices : visible;
    if (parts > total) {
        parts = total;
    }
    ddn_node* n = new (std::nothrow) ddn_node();
    if (!n) {
        return DDN_ENOMEM;
    }
    n->cfg = *cfg;
    n->sample_bytes = cfg->input_format == DDN_IN_CF32 ? 8 : 2;
    if (kind == DDN_NODE_FSK4) {
        n->fsk4_copy = *cfg->fsk4;
        n->cfg.fsk4 = &n->fsk4_copy;
    }
    if (kind == DDN_NODE_P25P2) {
        n->p25p2_copy = *cfg->p25p2;
        n->cfg.p25p2 = &n->p25p2_copy;
    }
    for (int r = 0; r < parts; r++) {
        Part* p = new (std::nothrow) Part();
        if (!p) {
            ddn_node_destroy(n);
            return DDN_ENOMEM;
        }
        p->device = r % visible;
        p->sample_bytes = n->sample_bytes;
        p->samples = cfg->samples_per_call;
        memset(&p->ccfg, 0, sizeof(p->ccfg));
        memset(&p->mcfg, 0, sizeof(p->mcfg));
        memset(&p->fcfg, 0, sizeof(p->fcfg));
        memset(&p->p2cfg, 0, sizeof(p->p2cfg));
        const size_t row = (size_t)cfg->samples_per_call * n->sample_bytes;
        if (kind == DDN_NODE_MIXED) {
            (void)ddn_mixed_partition(cfg->n_channels, cfg->n_dmr, cfg->n_nxdn48, r, parts, p->first3, p->count3);
            // (its block of the global channel index [P25 | DMR | NXDN48]: contiguous, so it starts in the first group it has channels of)
            const int g0[3] = {0, cfg->n_channels, cfg->n_channels + cfg->n_dmr};
            p->count = p->count3[0] + p->count3[1] + p->count3[2];
            p->first = 0;
            for (int g = 2; g >= 0; g--) {
                if (p->count3[g] > 0) {
                    p->first = g0[g] + p->first3[g];
                }
            }
            p->mcfg.n_p25 = p->count3[0];
            p->mcfg.n_dmr = p->count3[1];
            p->mcfg.n_nxdn48 = p->count3[2];
            p->mcfg.samples_per_call = cfg->samples_per_call;
            p->mcfg.block_len = cfg->block_len;
            p->mcfg.input_format = cfg->input_format;
            p->mcfg.vocoder = cfg->vocoder;
            p->mcfg.overlap = cfg->overlap;
            for (int g = 0; g < 3; g++) {
                p->in_bytes[g] = (size_t)p->count3[g] * row;
            }
            p->ops = &kMixed;
        } else {
            (void)ddn_node_partition(cfg->n_channels, r, parts, &p->first, &p->count);
            p->first3[0] = p->first;
            p->count3[0] = p->count;
            p->in_bytes[0] = (size_t)p->count * row;
            if (kind == DDN_NODE_P25) {
                p->ccfg.n_channels = p->count;
                p->ccfg.samples_per_call = cfg->samples_per_call;
                p->ccfg.block_len = cfg->block_len;
                p->ccfg.input_format = cfg->input_format;
                p->ccfg.vocoder = cfg->vocoder;
                p->ccfg.modulation = cfg->modulation;
                p->ops = &kP25;
            } else if (kind == DDN_NODE_FSK4) {
                p->fcfg = n->fsk4_copy;
                p->fcfg.n_channels = p->count;
                p->fcfg.samples_per_call = cfg->samples_per_call;
                p->fcfg.block_len = cfg->block_len;
                p->fcfg.input_format = cfg->input_format;
                p->fcfg.vocoder = cfg->vocoder;
                p->ops = &kFsk4;
            } else {
                p->p2cfg = n->p25p2_copy;
                p->p2cfg.n_channels = p->count;
                p->p2cfg.samples_per_call = cfg->samples_per_call;
                p->p2cfg.block_len = cfg->block_len;
                p->p2cfg.input_format = cfg->input_format;
                p->p2cfg.vocoder = cfg->vocoder;
                p->p2seed = cfg->p25p2_seed44;
                p->ops = &kP25p2;
            }
        }
        n->parts.push_back(p);
        try {
            p->th = std::thread(part_main, p);
        } catch (const std::system_error&) { // (no C++ exception crosses the C boundary)
            ddn_set_error("ddn_node_create: cannot start the host thread of part %d", r);
            ddn_node_destroy(n);
            return DDN_ENOMEM;
        }
    }
    const int rc = all(n, C_CREATE);
    if (rc != DDN_OK) {
        ddn_node_destroy(n);
        return rc;
    }
    *out = n;
    return DDN_OK;
}

extern "C" int
ddn_node_parts(const ddn_node* n) {
    return n ? (int)n->parts.size() : 0;
}

extern "C" int
ddn_node_kind_of(const ddn_node* n) {
    return n ? n->cfg.kind : DDN_EINVAL;
}

extern "C" int
ddn_node_part_info(const ddn_node* n, int part, int* device, int* first_channel, int* n_channels) {
    if (!n || part < 0 || part >= (int)n->parts.size()) {
        return DDN_EINVAL;
    }
    const Part* p = n->parts[(size_t)part];
    if (device) {
        *device = p->device;
    }
    if (first_channel) {
        *first_channel = p->first;
    }
    if (n_channels) {
        *n_channels = p->count;
    }
    return DDN_OK;
}

extern "C" int
ddn_node_part_groups(const ddn_node* n, int part, int32_t first3[3], int32_t count3[3]) {
    if (!n || part < 0 || part >= (int)n->parts.size() || !first3 || !count3) {
        return DDN_EINVAL;
    }
    const Part* p = n->parts[(size_t)part];
    for (int g = 0; g < 3; g++) {
        first3[g] = p->first3[g];
        count3[g] = p->count3[g];
    }
    return DDN_OK;
}

extern "C" ddn_p25_chain*
ddn_node_chain(ddn_node* n, int part) {
    return (n && n->cfg.kind == DDN_NODE_P25 && part >= 0 && part < (int)n->parts.size()) ? (ddn_p25_chain*)n->parts[(size_t)part]->chain
                                                                                          : nullptr;
}

extern "C" void*
ddn_node_chain_object(ddn_node* n, int part) {
    return (n && part >= 0 && part < (int)n->parts.size()) ? n->parts[(size_t)part]->chain : nullptr;
}

extern "C" int
ddn_node_run_host(ddn_node* n, const void* h_iq, const ddn_p25_chain_host_out* outs) {
    if (!n || !h_iq) {
        return DDN_EINVAL;
    }
    const size_t row = (size_t)n->cfg.samples_per_call * n->sample_bytes;
    LockAll lk(n);
    // h_iq: every channel's row in the order of the global channel index - for the mixed kind [P25 | DMR | NXDN48]
    const size_t group0[3] = {0, (size_t)n->cfg.n_channels, (size_t)n->cfg.n_channels + (size_t)n->cfg.n_dmr};
    for (size_t k = 0; k < n->parts.size(); k++) {
        Part* p = n->parts[k];
        for (int g = 0; g < 3; g++) {
            const size_t ch = (n->cfg.kind == DDN_NODE_MIXED ? group0[g] : 0) + (size_t)p->first3[g];
            p->a_iq3[g] = (g == 0 || n->cfg.kind == DDN_NODE_MIXED) && p->count3[g] > 0 ? (const uint8_t*)h_iq + ch * row : nullptr;
        }
        p->a_out = (outs && n->cfg.kind == DDN_NODE_P25) ? &outs[k] : nullptr;
    }
    return all_locked(n, C_RUN_HOST);
}

extern "C" int
ddn_node_run_device(ddn_node* n, const void* const* d_iq) {
    if (!n || !d_iq) {
        return DDN_EINVAL;
    }
    const bool mixed = n->cfg.kind == DDN_NODE_MIXED;
    LockAll lk(n);
    for (size_t k = 0; k < n->parts.size(); k++) {
        Part* p = n->parts[k];
        for (int g = 0; g < 3; g++) {
            p->a_iq3[g] = mixed ? d_iq[3 * k + (size_t)g] : (g == 0 ? d_iq[k] : nullptr);
            if ((g == 0 || mixed) && p->count3[g] > 0 && !p->a_iq3[g]) {
                return DDN_EINVAL;
            }
        }
    }
    return all_locked(n, C_RUN_DEV);
}

extern "C" int
ddn_node_wait(ddn_node* n) {
    return n ? all(n, C_WAIT) : DDN_EINVAL;
}

extern "C" int
ddn_node_flush(ddn_node* n) {
    return n ? all(n, C_FLUSH) : DDN_EINVAL;
}

static int
one_locked(ddn_node* n, int part, Cmd c) {
    Part* p = n->parts[(size_t)part];
    post(p, c);
    const int rc = join(p);
    if (rc != DDN_OK) {
        ddn_set_error("ddn_node: device %d: %s", p->device, p->err);
    }
    return rc;
}

extern "C" int
ddn_node_on_part(ddn_node* n, int part, int (*fn)(void* chain_object, void* arg), void* arg) {
    if (!n || !fn || part < 0 || part >= (int)n->parts.size()) {
        return DDN_EINVAL;
    }
    Part* p = n->parts[(size_t)part];
    std::lock_guard<std::mutex> lk(p->call_mu);
    p->a_fn = fn;
    p->a_arg = arg;
    return one_locked(n, part, C_CALL);
}

extern "C" int
ddn_node_device_alloc(ddn_node* n, int part, size_t bytes, void** out) {
    if (!n || !out || part < 0 || part >= (int)n->parts.size()) {
        return DDN_EINVAL;
    }
    Part* p = n->parts[(size_t)part];
    std::lock_guard<std::mutex> lk(p->call_mu);
    p->a_bytes = bytes;
    p->a_pp = out;
    return one_locked(n, part, C_ALLOC);
}

extern "C" int
ddn_node_device_upload(ddn_node* n, int part, void* d_dst, const void* h_src, size_t bytes) {
    if (!n || part < 0 || part >= (int)n->parts.size()) {
        return DDN_EINVAL;
    }
    Part* p = n->parts[(size_t)part];
    std::lock_guard<std::mutex> lk(p->call_mu);
    p->a_dst = d_dst;
    p->a_src = h_src;
    p->a_bytes = bytes;
    return one_locked(n, part, C_UPLOAD);
}

extern "C" int
ddn_node_device_download(ddn_node* n, int part, void* h_dst, const void* d_src, size_t bytes) {
    if (!n || part < 0 || part >= (int)n->parts.size()) {
        return DDN_EINVAL;
    }
    Part* p = n->parts[(size_t)part];
    std::lock_guard<std::mutex> lk(p->call_mu);
    p->a_dst = h_dst;
    p->a_src = d_src;
    p->a_bytes = bytes;
    return one_locked(n, part, C_DOWNLOAD);
}

extern "C" void
ddn_node_device_free(ddn_node* n, int part, void* ptr) {
    if (!n || part < 0 || part >= (int)n->parts.size()) {
        return;
    }
    Part* p = n->parts[(size_t)part];
    std::lock_guard<std::mutex> lk(p->call_mu);
    p->a_dst = ptr;
    (void)one_locked(n, part, C_FREE);
}

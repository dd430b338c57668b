// ddn_api_node.cpp - all the GPUs of one node from C (include/ddn_node.h): block partition of the channel index, one P25 chain object
// and one host thread per device.  Host-only code; no collective on the data path.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "ddn_hip.h"
#include "ddn_internal.h"
#include "ddn_node.h"

namespace {
enum Cmd { C_NONE = 0, C_CREATE, C_RUN_HOST, C_RUN_DEV, C_WAIT, C_FLUSH, C_ALLOC, C_UPLOAD, C_DOWNLOAD, C_FREE, C_QUIT };

struct Part {
    int device = 0, first = 0, count = 0;
    ddn_p25_chain* chain = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    Cmd cmd = C_NONE;
    bool done = true;
    int rc = DDN_OK;
    // arguments of the pending command
    const void* a_iq = nullptr;
    const ddn_p25_chain_host_out* a_out = nullptr;
    size_t a_bytes = 0;
    void* a_dst = nullptr;
    const void* a_src = nullptr;
    void** a_pp = nullptr;
    ddn_p25_chain_config ccfg;
    char err[256] = {0};
};
} // namespace

struct ddn_node {
    ddn_node_config cfg;
    std::vector<Part*> parts;
    size_t sample_bytes;
};

static int
hip_rc(hipError_t e, const char* what) {
    if (e == hipSuccess) {
        return DDN_OK;
    }
    ddn_set_error("%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? DDN_ENOMEM : DDN_EHIP;
}

static void
part_main(Part* p) {
    (void)hipSetDevice(p->device);
    for (;;) {
        Cmd c;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return !p->done; });
            c = p->cmd;
        }
        int rc = DDN_OK;
        switch (c) {
            case C_CREATE:
                rc = ddn_p25_chain_create(&p->ccfg, &p->chain);
                if (rc == DDN_OK) {
                    rc = ddn_p25_chain_set_first_channel(p->chain, p->first);
                }
                break;
            case C_RUN_HOST: rc = ddn_p25_chain_run_host(p->chain, p->a_iq, p->a_out); break;
            case C_RUN_DEV: rc = ddn_p25_chain_run_pipelined(p->chain, p->a_iq); break;
            case C_WAIT: rc = ddn_p25_chain_wait(p->chain); break;
            case C_FLUSH: rc = ddn_p25_chain_flush(p->chain); break;
            case C_ALLOC: rc = hip_rc(hipMalloc(p->a_pp, p->a_bytes), "hipMalloc"); break;
            case C_UPLOAD: rc = hip_rc(hipMemcpy(p->a_dst, p->a_src, p->a_bytes, hipMemcpyHostToDevice), "hipMemcpy H2D"); break;
            case C_DOWNLOAD: rc = hip_rc(hipMemcpy(p->a_dst, p->a_src, p->a_bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H"); break;
            case C_FREE: (void)hipFree(p->a_dst); break;
            case C_QUIT:
                ddn_p25_chain_destroy(p->chain);
                p->chain = nullptr;
                break;
            default: break;
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

static int
all(ddn_node* n, Cmd c) {
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

extern "C" void
ddn_node_destroy(ddn_node* n) {
    if (!n) {
        return;
    }
    for (Part* p : n->parts) {
        if (p->th.joinable()) {
            post(p, C_QUIT);
            (void)join(p);
            p->th.join();
        }
        delete p;
    }
    delete n;
}

extern "C" int
ddn_node_create(const ddn_node_config* cfg, ddn_node** out) {
    if (!cfg || !out || cfg->n_channels <= 0 || cfg->samples_per_call <= 0 || cfg->block_len <= 0 || cfg->n_devices < 0) {
        ddn_set_error("ddn_node_create: bad configuration");
        return DDN_EINVAL;
    }
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        ddn_set_error("ddn_node_create: no HIP device");
        return DDN_ENODEV;
    }
    int parts = cfg->n_devices > 0 ? cfg->n_devices : visible;
    if (parts > cfg->n_channels) {
        parts = cfg->n_channels;
    }
    ddn_node* n = new (std::nothrow) ddn_node();
    if (!n) {
        return DDN_ENOMEM;
    }
    n->cfg = *cfg;
    n->sample_bytes = cfg->input_format == DDN_IN_CF32 ? 8 : 2;
    for (int r = 0; r < parts; r++) {
        Part* p = new (std::nothrow) Part();
        if (!p) {
            ddn_node_destroy(n);
            return DDN_ENOMEM;
        }
        p->device = r % visible;
        (void)ddn_node_partition(cfg->n_channels, r, parts, &p->first, &p->count);
        memset(&p->ccfg, 0, sizeof(p->ccfg));
        p->ccfg.n_channels = p->count;
        p->ccfg.samples_per_call = cfg->samples_per_call;
        p->ccfg.block_len = cfg->block_len;
        p->ccfg.input_format = cfg->input_format;
        p->ccfg.vocoder = cfg->vocoder;
        p->ccfg.modulation = cfg->modulation;
        n->parts.push_back(p);
        p->th = std::thread(part_main, p);
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

extern "C" ddn_p25_chain*
ddn_node_chain(ddn_node* n, int part) {
    return (n && part >= 0 && part < (int)n->parts.size()) ? n->parts[(size_t)part]->chain : nullptr;
}

extern "C" int
ddn_node_run_host(ddn_node* n, const void* h_iq, const ddn_p25_chain_host_out* outs) {
    if (!n || !h_iq) {
        return DDN_EINVAL;
    }
    for (size_t k = 0; k < n->parts.size(); k++) {
        Part* p = n->parts[k];
        p->a_iq = (const uint8_t*)h_iq + (size_t)p->first * (size_t)n->cfg.samples_per_call * n->sample_bytes;
        p->a_out = outs ? &outs[k] : nullptr;
    }
    return all(n, C_RUN_HOST);
}

extern "C" int
ddn_node_run_device(ddn_node* n, const void* const* d_iq) {
    if (!n || !d_iq) {
        return DDN_EINVAL;
    }
    for (size_t k = 0; k < n->parts.size(); k++) {
        if (!d_iq[k]) {
            return DDN_EINVAL;
        }
        n->parts[k]->a_iq = d_iq[k];
    }
    return all(n, C_RUN_DEV);
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
one(ddn_node* n, int part, Cmd c) {
    Part* p = n->parts[(size_t)part];
    post(p, c);
    const int rc = join(p);
    if (rc != DDN_OK) {
        ddn_set_error("ddn_node: device %d: %s", p->device, p->err);
    }
    return rc;
}

extern "C" int
ddn_node_device_alloc(ddn_node* n, int part, size_t bytes, void** out) {
    if (!n || !out || part < 0 || part >= (int)n->parts.size()) {
        return DDN_EINVAL;
    }
    Part* p = n->parts[(size_t)part];
    p->a_bytes = bytes;
    p->a_pp = out;
    return one(n, part, C_ALLOC);
}

extern "C" int
ddn_node_device_upload(ddn_node* n, int part, void* d_dst, const void* h_src, size_t bytes) {
    if (!n || part < 0 || part >= (int)n->parts.size()) {
        return DDN_EINVAL;
    }
    Part* p = n->parts[(size_t)part];
    p->a_dst = d_dst;
    p->a_src = h_src;
    p->a_bytes = bytes;
    return one(n, part, C_UPLOAD);
}

extern "C" int
ddn_node_device_download(ddn_node* n, int part, void* h_dst, const void* d_src, size_t bytes) {
    if (!n || part < 0 || part >= (int)n->parts.size()) {
        return DDN_EINVAL;
    }
    Part* p = n->parts[(size_t)part];
    p->a_dst = h_dst;
    p->a_src = d_src;
    p->a_bytes = bytes;
    return one(n, part, C_DOWNLOAD);
}

extern "C" void
ddn_node_device_free(ddn_node* n, int part, void* ptr) {
    if (!n || part < 0 || part >= (int)n->parts.size()) {
        return;
    }
    n->parts[(size_t)part]->a_dst = ptr;
    (void)one(n, part, C_FREE);
}

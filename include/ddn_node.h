/* ddn_node.h - one node, all of its GPUs, from C (channels are independent streams, so the channel index is block-
 * partitioned over the devices - the first n % D devices one channel more - and nothing crosses between them on the data path).
 *
 * A ddn_node owns one chain object (include/ddn_chain.h) per device - the P25 Phase 1 chain by default; since round 6 any of the
 * four chain objects, ddn_node_config.kind - and one host thread per device that makes that
 * device's calls with the device current: the C counterpart of what bench.py does with one process per GPU.  What it stands in for in a
 * dsd-neo host: N instances of the per-stream demodulator + processFrame() loop (src/io/radio/rtl_sdr_fm.cpp:3458-3516,
 * src/engine/protocol_dispatch.c:30-44), the host's own channel table deciding which capture goes to which of them.
 *
 *   ddn_node_config nc = { .n_channels = 32768, .samples_per_call = 48000, .block_len = 8192, .input_format = DDN_IN_CU8, .vocoder = 1 };
 *   ddn_node* node; ddn_node_create(&nc, &node);                      // 8 devices visible: 4096 channels each
 *   for (;;) { ddn_node_run_host(node, pinned_iq, outs); ... }        // pinned_iq [n_channels][samples][2]: device d takes its block
 *   ddn_node_wait(node); ddn_node_flush(node); ddn_node_destroy(node);
 */
#ifndef DDN_NODE_H
#define DDN_NODE_H
#include <stddef.h>
#include <stdint.h>

#include "ddn_chain.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ddn_node_config {
    int n_channels;       /* over all devices */
    int samples_per_call;
    int block_len;
    int input_format;     /* DDN_IN_CU8 / DDN_IN_CF32 */
    int vocoder;
    int modulation;       /* DDN_P25_MOD_* */
    int n_devices;        /* 0 = every visible device; more than are visible: the list wraps around (several chain objects share a
                             device - for tests on a one-GPU box, never a production layout) */
    /* appended in round 6 (all zero = the P25 Phase 1 node as before): */
    int kind;             /* DDN_NODE_P25 / _MIXED / _FSK4 / _P25P2: which chain object a part owns */
    int n_dmr, n_nxdn48;  /* MIXED (BASELINE configs[3]): channels of the other two groups; n_channels is the P25 group.  Every group
                             is block-partitioned over the parts (ddn_mixed_partition), a part owns one ddn_mixed_chain */
    int overlap;          /* MIXED: ddn_mixed_chain_config.overlap */
    const ddn_fsk4_chain_config* fsk4;   /* FSK4: protocol, rf_mod, inverted, handlers of every part's ddn_fsk4_chain (its n_channels /
                                            samples_per_call / block_len / input_format / vocoder are replaced by the part's) */
    const ddn_p25p2_chain_config* p25p2; /* P25P2: sample_rate_hz, max_groups, snr_cqpsk_db of every part's ddn_p25p2_chain */
    const uint64_t* p25p2_seed44;        /* P25P2: [n_channels] scrambler seeds (WACN << 24 | SYS << 12 | NAC), NULL = as ddn_p25p2_chain_create
                                            without seeds; must stay valid until ddn_node_create returns */
} ddn_node_config;
enum { DDN_NODE_P25 = 0, DDN_NODE_MIXED = 1, DDN_NODE_FSK4 = 2, DDN_NODE_P25P2 = 3 };
typedef struct ddn_node ddn_node;

/* block partition of [0, n_channels) over `world` parts: part `rank` owns [*first, *first + *count) */
int ddn_node_partition(int n_channels, int rank, int world, int* first, int* count);

int ddn_node_create(const ddn_node_config* cfg, ddn_node** out);
void ddn_node_destroy(ddn_node* n);
int ddn_node_parts(const ddn_node* n);                                  /* chain objects = worker threads */
int ddn_node_part_info(const ddn_node* n, int part, int* device, int* first_channel, int* n_channels);
ddn_p25_chain* ddn_node_chain(ddn_node* n, int part);                    /* kind P25 (NULL otherwise); results: ddn_p25_chain_get_results() with the part's device current */
int ddn_node_kind_of(const ddn_node* n);
/* a part's chain object whatever the kind: ddn_p25_chain* / ddn_mixed_chain* / ddn_fsk4_chain* / ddn_p25p2_chain* */
void* ddn_node_chain_object(ddn_node* n, int part);
/* MIXED: the part's block of each group, first3[g] / count3[g] for g = P25, DMR, NXDN48 (other kinds: group 0 = the part's block) */
int ddn_node_part_groups(const ddn_node* n, int part, int32_t first3[3], int32_t count3[3]);
/* fn(chain object, arg) on the part's host thread, its device current (result getters, timing switches ...); returns fn's result */
int ddn_node_on_part(ddn_node* n, int part, int (*fn)(void* chain_object, void* arg), void* arg);

/* One step on every device: part p runs ddn_p25_chain_run_host() on its block of h_iq (pinned host memory, [n_channels][samples][2] u8 or
 * [..][2] f32) with outs[p] (NULL: no result copies; an entry's pointers address THAT part's arrays).  Returns when every part's call has
 * returned (i.e. everything is queued and the previous input buffer may be refilled - ddn_p25_chain_run_host's contract, per part);
 * the first error of any part is the result.
 * Kinds other than P25: h_iq holds every channel's row in the order of the global channel index (MIXED: the P25 group's rows, then the
 * DMR group's, then the NXDN48 group's); a part copies its blocks into device buffers of its own (two sets, used in turn) and runs its
 * chain object on them; outs is ignored - the results stay on the device (ddn_node_chain_object + the kind's _get_results, through
 * ddn_node_on_part where the caller's thread has another device current). */
int ddn_node_run_host(ddn_node* n, const void* h_iq, const ddn_p25_chain_host_out* outs);
/* the same with device-resident input: d_iq[p] = that part's I/Q on its device (ddn_p25_chain_run_pipelined; MIXED: d_iq[3 p + g] =
 * part p's group g, NULL where the part has no channel of the group - ddn_mixed_chain_run) */
int ddn_node_run_device(ddn_node* n, const void* const* d_iq);
int ddn_node_wait(ddn_node* n);
int ddn_node_flush(ddn_node* n);
/* device memory on a part's device (for hosts without a HIP binding of their own) */
int ddn_node_device_alloc(ddn_node* n, int part, size_t bytes, void** out);
int ddn_node_device_upload(ddn_node* n, int part, void* d_dst, const void* h_src, size_t bytes);
int ddn_node_device_download(ddn_node* n, int part, void* h_dst, const void* d_src, size_t bytes);
void ddn_node_device_free(ddn_node* n, int part, void* p);

#ifdef __cplusplus
}
#endif
#endif

/* ddn_node.h - one node, all of its GPUs, from C (channels are independent streams, so the channel index is block-
 * partitioned over the devices - the first n % D devices one channel more - and nothing crosses between them on the data path).
 *
 * A ddn_node owns one P25 Phase 1 chain object (include/ddn_chain.h) per device and one host thread per device that makes that
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
} ddn_node_config;
typedef struct ddn_node ddn_node;

/* block partition of [0, n_channels) over `world` parts: part `rank` owns [*first, *first + *count) */
int ddn_node_partition(int n_channels, int rank, int world, int* first, int* count);

int ddn_node_create(const ddn_node_config* cfg, ddn_node** out);
void ddn_node_destroy(ddn_node* n);
int ddn_node_parts(const ddn_node* n);                                  /* chain objects = worker threads */
int ddn_node_part_info(const ddn_node* n, int part, int* device, int* first_channel, int* n_channels);
ddn_p25_chain* ddn_node_chain(ddn_node* n, int part);                    /* results: ddn_p25_chain_get_results() with the part's device current */

/* One step on every device: part p runs ddn_p25_chain_run_host() on its block of h_iq (pinned host memory, [n_channels][samples][2] u8 or
 * [..][2] f32) with outs[p] (NULL: no result copies; an entry's pointers address THAT part's arrays).  Returns when every part's call has
 * returned (i.e. everything is queued and the previous input buffer may be refilled - ddn_p25_chain_run_host's contract, per part);
 * the first error of any part is the result. */
int ddn_node_run_host(ddn_node* n, const void* h_iq, const ddn_p25_chain_host_out* outs);
/* the same with device-resident input: d_iq[p] = that part's I/Q on its device (ddn_p25_chain_run_pipelined) */
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

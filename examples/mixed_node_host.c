/* examples/mixed_node_host.c - BASELINE configs[3] from one C program (include/ddn_node.h, kind = DDN_NODE_MIXED): a mixed batch of
 * P25 Phase 1, DMR and NXDN48 channels block-partitioned over every GPU of the node (ddn_mixed_partition gives each device its block
 * of each protocol), one ddn_mixed_chain + one host thread per device, input from pinned host memory in the order of the global
 * channel index [P25 | DMR | NXDN48].  Results stay on the devices; a getter runs on the part's own thread (ddn_node_on_part).
 *   gcc -std=c11 -I include examples/mixed_node_host.c -L dsd-neo_amd -ldsdneo_hip -Wl,-rpath,$PWD/dsd-neo_amd -o mixed_node_host
 *   ./mixed_node_host [channels] [steps] [parts (0 = every visible device)]
 * tests/test_cabi_exports.py compiles and links it; tests/test_node_gpu.py runs it on the GPU box. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <ddn_node.h>

static double
now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

/* runs on the part's host thread, that part's device current: how many records the part's DMR group holds after the last call */
static int
count_dmr_records(void* chain_object, void* arg) {
    ddn_fsk4_chain* dmr = (ddn_fsk4_chain*)ddn_mixed_chain_part((ddn_mixed_chain*)chain_object, 1);
    long* total = (long*)arg;
    if (!dmr) {
        return DDN_OK; /* this part has no DMR channel */
    }
    ddn_fsk4_chain_results r;
    int rc = ddn_fsk4_chain_get_results(dmr, &r);
    if (rc != DDN_OK) {
        return rc;
    }
    int32_t cnt[4096];
    const int n = (int)total[1] < 4096 ? (int)total[1] : 4096;
    rc = ddn_device_download(cnt, r.d_counts, sizeof(int32_t) * (size_t)n);
    for (int c = 0; rc == DDN_OK && c < n; c++) {
        total[0] += cnt[c];
    }
    return rc;
}

int
main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 768, n = 48000, steps = argc > 2 ? atoi(argv[2]) : 4;
    const int third = B / 3, Bp = B - 2 * third, Bd = third, Bn = third;
    ddn_node_config nc = {.n_channels = Bp, .samples_per_call = n, .block_len = 8192, .input_format = DDN_IN_CU8, .vocoder = 1,
                          .n_devices = argc > 3 ? atoi(argv[3]) : 0, .kind = DDN_NODE_MIXED, .n_dmr = Bd, .n_nxdn48 = Bn};
    ddn_node* node;
    if (ddn_node_create(&nc, &node) != DDN_OK) {
        fprintf(stderr, "%s\n", ddn_last_error());
        return 1;
    }
    const size_t bytes = (size_t)B * n * 2;
    void* iq[2];
    for (int k = 0; k < 2; k++) { /* two input buffers: one being copied, one being refilled */
        if (ddn_host_alloc_pinned(bytes, &iq[k]) != DDN_OK) {
            fprintf(stderr, "%s\n", ddn_last_error());
            return 1;
        }
        memset(iq[k], 127, bytes);
    }
    for (int p = 0; p < ddn_node_parts(node); p++) {
        int dev;
        int32_t first3[3], count3[3];
        ddn_node_part_info(node, p, &dev, NULL, NULL);
        ddn_node_part_groups(node, p, first3, count3);
        printf("part %d: device %d, P25 %d..%d, DMR %d..%d, NXDN48 %d..%d\n", p, dev, first3[0], first3[0] + count3[0] - 1, first3[1],
               first3[1] + count3[1] - 1, first3[2], first3[2] + count3[2] - 1);
    }
    const double t0 = now_ms();
    for (int s = 0; s < steps; s++) {
        if (ddn_node_run_host(node, iq[s & 1], NULL) != DDN_OK) {
            fprintf(stderr, "%s\n", ddn_last_error());
            return 1;
        }
    }
    if (ddn_node_wait(node) != DDN_OK) {
        fprintf(stderr, "%s\n", ddn_last_error());
        return 1;
    }
    const double ms = (now_ms() - t0) / steps;
    printf("%d mixed channels (%d P25 + %d DMR + %d NXDN48) x %d samples per step: %.2f ms per step, %.2f Gsamples/s from host memory\n", B, Bp,
           Bd, Bn, n, ms, (double)B * n / ms * 1e-6);
    long records = 0;
    for (int p = 0; p < ddn_node_parts(node); p++) {
        int32_t first3[3], count3[3];
        ddn_node_part_groups(node, p, first3, count3);
        long arg[2] = {0, count3[1]};
        if (ddn_node_on_part(node, p, count_dmr_records, arg) != DDN_OK) {
            fprintf(stderr, "%s\n", ddn_last_error());
            return 1;
        }
        records += arg[0];
    }
    printf("DMR records held after the last call: %ld\n", records);
    ddn_node_flush(node);
    ddn_node_destroy(node);
    ddn_host_free_pinned(iq[0]);
    ddn_host_free_pinned(iq[1]);
    return 0;
}

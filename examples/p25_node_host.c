/* examples/p25_node_host.c - every GPU of the node from one C program (include/ddn_node.h): B P25 Phase 1 channels of cu8 I/Q, block
 * partitioned over the visible devices, one ddn_p25_chain + one host thread per device, input from pinned host memory.
 *   gcc -std=c11 -I include examples/p25_node_host.c -L dsd-neo_amd -ldsdneo_hip -Wl,-rpath,$PWD/dsd-neo_amd -o p25_node_host
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

int
main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, n = 48000, steps = argc > 2 ? atoi(argv[2]) : 4;
    ddn_node_config nc = {.n_channels = B, .samples_per_call = n, .block_len = 8192, .input_format = DDN_IN_CU8, .vocoder = 1,
                          .n_devices = argc > 3 ? atoi(argv[3]) : 0};
    ddn_node* node;
    if (ddn_node_create(&nc, &node) != DDN_OK) {
        fprintf(stderr, "%s\n", ddn_last_error());
        return 1;
    }
    const size_t bytes = (size_t)B * n * 2;
    void* iq[2];
    for (int k = 0; k < 2; k++) { /* two input buffers: one being read by the copy engines, one being refilled */
        if (ddn_host_alloc_pinned(bytes, &iq[k]) != DDN_OK) {
            fprintf(stderr, "%s\n", ddn_last_error());
            return 1;
        }
        memset(iq[k], 127, bytes);
    }
    for (int p = 0; p < ddn_node_parts(node); p++) {
        int dev, first, count;
        ddn_node_part_info(node, p, &dev, &first, &count);
        printf("part %d: device %d, channels %d..%d\n", p, dev, first, first + count - 1);
    }
    const double t0 = now_ms();
    for (int s = 0; s < steps; s++) {
        if (ddn_node_run_host(node, iq[s & 1], NULL) != DDN_OK) {
            fprintf(stderr, "%s\n", ddn_last_error());
            return 1;
        }
    }
    ddn_node_wait(node);
    const double ms = (now_ms() - t0) / steps;
    printf("%d channels x %d samples per step: %.2f ms per step, %.2f Gsamples/s from host memory\n", B, n, ms, (double)B * n / ms * 1e-6);
    ddn_node_flush(node);
    ddn_node_destroy(node);
    ddn_host_free_pinned(iq[0]);
    ddn_host_free_pinned(iq[1]);
    return 0;
}

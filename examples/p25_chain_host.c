/* examples/p25_chain_host.c - INTEGRATION.md's fifteen-line host, as a program: B P25 Phase 1 channels of cu8 I/Q from a file
 * ([second][channel][sample][I,Q]; zeros when no file is given) through ddn_p25_chain, one call per second of air time.
 *   gcc -std=c11 -I include examples/p25_chain_host.c -L dsd-neo_amd -ldsdneo_hip -Wl,-rpath,$PWD/dsd-neo_amd -o p25_chain_host
 * tests/test_cabi_exports.py compiles and links it (running it needs an MI355X). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <ddn_chain.h>

int
main(int argc, char** argv) {
    const int B = 64, n = 48000;
    ddn_p25_chain_config cfg = {.n_channels = B, .samples_per_call = n, .block_len = 8192, .input_format = DDN_IN_CU8, .vocoder = 1};
    ddn_p25_chain* ch;
    if (ddn_p25_chain_create(&cfg, &ch) != DDN_OK) {
        fprintf(stderr, "%s\n", ddn_last_error());
        return 1;
    }
    const size_t bytes = (size_t)B * n * 2;
    uint8_t* host_iq = malloc(bytes);
    void* d_iq;
    FILE* f = argc > 1 ? fopen(argv[1], "rb") : NULL;
    if (!host_iq || ddn_device_alloc(bytes, &d_iq) != DDN_OK) {
        return 1;
    }
    memset(host_iq, 127, bytes);
    for (int second = 0; second < 3 && (!f || fread(host_iq, 1, bytes, f) == bytes); second++) {
        ddn_device_upload(d_iq, host_iq, bytes);
        ddn_p25_chain_run_pipelined(ch, d_iq); /* front end -> loop + handlers -> framer -> FEC -> IMBE -> PCM */
        ddn_p25_chain_wait(ch);
        ddn_p25_chain_results r;
        ddn_p25_chain_get_results(ch, &r);
        int32_t n_syncs[64];
        ddn_device_download(n_syncs, r.d_n_syncs, sizeof(n_syncs));
        printf("second %d: channel 0 decoded %d frames\n", second, n_syncs[0]);
    }
    ddn_p25_chain_flush(ch); /* the frames the carry still held back */
    ddn_p25_chain_destroy(ch);
    ddn_device_free(d_iq);
    free(host_iq);
    return 0;
}

/* ddn_internal.h — shared between the C-ABI translation units of libdsdneo_hip.so (not installed). */
#ifndef DDN_INTERNAL_H
#define DDN_INTERNAL_H

#include "../../include/ddn_hip.h"

#define DDN_MAX_TAPS   143 /* odd; reference kChannelLpfTaps = 144 (src/dsp/demod_pipeline.cpp:129) */
#define DDN_MAX_CENTER 71

#ifdef __cplusplus
extern "C" {
#endif
int ddn_design_channel_lpf(int rate_hz, int profile, float* taps, int max_taps);
#define DDN_FLL_MAX_TAPS 48 /* FLL_BAND_EDGE_MAX_TAPS, include/dsd-neo/dsp/costas.h:60 */
int ddn_design_fll_band_edge(int sps, float* taps4, float* alpha, float* beta);
int ddn_design_resampler(int L, int M, float* taps);
#define DDN_RESAMP_MAX_L 512
int ddn_p25p1_layout_nid(int32_t out32[32]);
int ddn_p25p1_layout_trellis_block(int block, int32_t out98[98]);
int ddn_p25p1_layout_ldu_words(int ldu, int32_t out120[120]);
int ddn_p25p1_layout_ldu_imbe(int32_t first9[9], int32_t status9[9]);
int ddn_p25p1_layout_hdu(int32_t hex3[36 * 3], int32_t par6[36 * 6]);
int ddn_p25p1_layout_tdulc(int32_t data6[72], int32_t par6[72]);
int ddn_p25p1_layout_ldu_lsd(int32_t out16[16]);
void ddn_set_error(const char* fmt, ...);
/* the next ddn_p25_rx_run() records this HIP event between its matched filter and its loop kernel (one shot) */
int ddn_p25_rx_mark_loop_start(ddn_p25_rx* b, void* hip_event);
/* the next ddn_p25_rx_run() makes its stream wait for this HIP event between its matched filter and its loop kernel (one shot) */
int ddn_p25_rx_gate_loop(ddn_p25_rx* b, void* hip_event);
/* the mixed chain's shared front end: a part's stage 0 without its own front-end launch, and where that launch has to write */
struct ddn_p25_chain;
struct ddn_fsk4_chain;
int ddn_p25_chain_stage0_prepare(struct ddn_p25_chain* c, void* hip_stream, float** disc_out);
float* ddn_fsk4_chain_disc_buffer(struct ddn_fsk4_chain* c);
void* ddn_fsk4_chain_reads_done_event(struct ddn_fsk4_chain* c);
/* the two kernels of ddn_mbe_synth_batch as separate calls (ddn_api_mbe.cpp) */
struct ddn_mbe_batch;
int ddn_mbe_params_only(struct ddn_mbe_batch* b, const uint8_t* d_bits, const int32_t* d_result_in, size_t n_frames, int32_t* d_result_out,
                        void* hip_stream);
int ddn_mbe_synth_only(struct ddn_mbe_batch* b, size_t n_frames, float* d_pcm, void* hip_stream);
#ifdef __cplusplus
}
#endif
#endif

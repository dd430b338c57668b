// ddn_p25p2_seq.h - launchers of ddn_p25p2_seq.hip (P25 Phase 2 sequencing above the burst layer)
#pragma once
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdint.h>

#include "ddn_hip.h"

hipError_t ddn_dev_p2_rows(const uint8_t* bits1400, const int16_t* llr1400, size_t n_groups_total, uint8_t* rb, int16_t* rl, hipStream_t st);
hipError_t ddn_dev_p2_voice_gather(const int32_t* info, const int32_t* groups_of, int n_channels, int n_groups, int cap, const uint8_t* fr,
                                   const uint8_t* rel, int32_t* src, int32_t* count, uint8_t* o_fr, uint8_t* o_rel, uint8_t* skip, hipStream_t st);
hipError_t ddn_dev_p2_cut_records(const uint8_t* rec, size_t stride, const int32_t* sync_pos, const int32_t* n_sync, int n_channels, int max_groups,
                                  uint8_t* bits1400, int16_t* llr1400, hipStream_t st);
hipError_t ddn_dev_p2_sequence(const int32_t* duid, const int32_t* isch, int n_channels, int n_groups, const int32_t* groups_of,
                               const uint64_t* seed44,
                               ddn_p25p2_seq_state* state, int32_t* info, int32_t* row_off, int32_t* seq_of, int32_t* counts, int32_t* list,
                               int32_t* ess_src, int32_t* final_src, hipStream_t st);
hipError_t ddn_dev_p2_gather(int cls, int count, const int32_t* list, const int32_t* info, const uint8_t* rb, const int16_t* rl, const uint8_t* xb,
                             const int16_t* xl, uint8_t* db, int16_t* dl, const int32_t* ess_src, const ddn_p25p2_seq_state* state,
                             int rows_per_channel, uint8_t* ess_pl, int16_t* ess_pll, uint8_t* ess_pa, int16_t* ess_pal, hipStream_t st);
hipError_t ddn_dev_p2_state_ess(const int32_t* final_src, const uint8_t* xb, const int16_t* xl, int n_channels, ddn_p25p2_seq_state* state,
                                hipStream_t st);
hipError_t ddn_dev_p2_scatter(int cls, int count, const int32_t* list, int32_t* info, const uint8_t* x_payload, int n_pl, const int32_t* ec,
                              const uint8_t* used, const uint8_t* c12, const uint8_t* c16, const uint8_t* fr, const uint8_t* rel, int frame_count,
                              const uint8_t* ess_out, uint8_t* o_payload, uint8_t* o_fr, uint8_t* o_rel, uint8_t* o_ess, hipStream_t st);
hipError_t ddn_dev_p2_sync_cut(const uint8_t* dibits, const int16_t* llr2, int n_channels, int n, size_t stride, const int32_t* cursor_in,
                               int max_groups, int32_t* n_groups, int32_t* group_pos, int32_t* cursor_out, uint8_t* bits1400, int16_t* llr1400,
                               hipStream_t st);

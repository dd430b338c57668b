/* ddn_host_mode.c - the demod thread's mode matrix: which output kind, symbol rate, level count, channel
 * LPF profile and samples per symbol a set of enabled protocols gets, i.e. what rtl_demod_init_for_mode() +
 * demod_apply_channel_lpf_defaults() decide before the first full_demod() call.  Host logic only.
 *   reference: src/io/radio/rtl_demod_config.cpp:63-182 (mode counts, symbol rate, levels), :189-229 (channel profile),
 *              :231-258 (output kind), :491-509 (LPF defaults: on from 20 kHz, CQPSK output forces its profile), :535-553
 *              (cqpsk_enable = mod_qpsk); dsd_opts_uses_wide_4800_profile include/dsd-neo/core/opts.h:402-409;
 *              documented table docs/rtl-demod-pipeline-audit.md:36-51. */
#include "ddn_hip.h"

static int
mode_count(const ddn_mode_flags* f) {
    return (f->p25p1 == 1) + (f->p25p2 == 1) + (f->provoice == 1) + (f->dmr == 1) + (f->nxdn48 == 1) + (f->nxdn96 == 1)
           + (f->x2tdma == 1) + (f->ysf == 1) + (f->dstar == 1) + (f->dpmr == 1) + (f->m17 == 1);
}

static int
symbol_rate(const ddn_mode_flags* f) {
    const int n = mode_count(f);
    if (f->provoice == 1 && n == 1) {
        return 9600;
    }
    if ((f->p25p2 == 1 || f->x2tdma == 1) && f->p25p1 == 0 && n == (f->p25p2 == 1) + (f->x2tdma == 1)) {
        return 6000;
    }
    if ((f->nxdn48 == 1 || f->dpmr == 1) && n == (f->nxdn48 == 1) + (f->dpmr == 1)) {
        return 2400;
    }
    return 4800;
}

static int
any_four_level(const ddn_mode_flags* f) {
    return f->p25p1 == 1 || f->p25p2 == 1 || f->dmr == 1 || f->nxdn48 == 1 || f->nxdn96 == 1 || f->x2tdma == 1 || f->ysf == 1
           || f->dpmr == 1 || f->m17 == 1;
}

static int
levels_for_rate(const ddn_mode_flags* f, int rate) {
    const int four_4800 = f->p25p1 == 1 || f->dmr == 1 || f->nxdn96 == 1 || f->ysf == 1 || f->m17 == 1;
    if (rate == 9600 && f->provoice == 1) {
        return 2;
    }
    if (rate == 4800 && f->dstar == 1 && !four_4800) {
        return 2;
    }
    if ((f->dstar == 1 || f->provoice == 1) && !any_four_level(f)) {
        return 2;
    }
    return 4;
}

static int
channel_profile(const ddn_mode_flags* f, int cqpsk, int rate) {
    switch (rate) {
        case 9600:
            if (f->provoice == 1) {
                return DDN_LPF_PROVOICE;
            }
            break;
        case 2400:
            if (f->nxdn48 == 1 || f->dpmr == 1) {
                return DDN_LPF_6K25;
            }
            break;
        case 6000:
            if (f->p25p2 == 1 || f->x2tdma == 1) {
                return f->p25p2 == 1 ? DDN_LPF_P25_CQPSK : DDN_LPF_12K5;
            }
            break;
        default: break;
    }
    if (f->dmr == 1 || f->nxdn96 == 1 || f->ysf == 1 || f->m17 == 1) {
        return DDN_LPF_12K5;
    }
    if (f->p25p1 == 1 || f->p25p2 == 1) {
        return cqpsk ? DDN_LPF_P25_CQPSK : DDN_LPF_P25_C4FM;
    }
    if (f->nxdn48 == 1 || f->dpmr == 1 || f->dstar == 1) {
        return DDN_LPF_6K25;
    }
    if (f->x2tdma == 1) {
        return DDN_LPF_12K5;
    }
    if (f->provoice == 1) {
        return DDN_LPF_PROVOICE;
    }
    return DDN_LPF_WIDE;
}

int
ddn_mode_config(const ddn_mode_flags* flags, int demod_rate_hz, ddn_mode_result* out) {
    if (!flags || !out || demod_rate_hz <= 0) {
        return DDN_EINVAL;
    }
    const ddn_mode_flags* f = flags;
    int cqpsk = f->mod_qpsk == 1;
    out->symbol_rate_hz = symbol_rate(f);
    out->symbol_levels = levels_for_rate(f, out->symbol_rate_hz);
    if (mode_count(f) == 0 || f->analog_only == 1) {
        out->output_kind = DDN_OUTPUT_AUDIO_MONITOR;
    } else if (cqpsk) {
        out->output_kind = DDN_OUTPUT_SYMBOL_CQPSK;
        out->symbol_levels = 4;
    } else {
        out->output_kind = DDN_OUTPUT_FSK_DISCRIMINATOR;
    }
    if (out->output_kind == DDN_OUTPUT_FSK_DISCRIMINATOR) {
        cqpsk = 0;
    }
    out->cqpsk_enable = out->output_kind == DDN_OUTPUT_SYMBOL_CQPSK;
    out->ted_enabled = out->cqpsk_enable;
    out->channel_lpf_enable = demod_rate_hz >= 20000;
    out->lpf_profile = out->channel_lpf_enable ? channel_profile(f, cqpsk, out->symbol_rate_hz) : DDN_LPF_WIDE;
    if (out->output_kind == DDN_OUTPUT_SYMBOL_CQPSK) {
        out->lpf_profile = DDN_LPF_P25_CQPSK;
    }
    out->samples_per_symbol = (demod_rate_hz + out->symbol_rate_hz / 2) / out->symbol_rate_hz;
    return DDN_OK;
}

/* ddn_host_files.c - the two user-visible output files downstream of the path, host C:
 *   ddn_symbol_capture_write   dsd-neo's soft symbol-capture file (-c): 16-byte header "DSDNSYM2", version 2, record size 10
 *                              (openSymbolOutFile, src/core/file/dsd_file.c:876-890) + one 10-byte record per symbol
 *                              (write_symbol_capture_record, src/core/frames/dsd_dibit.c:794-818).  The receive loops' records
 *                              are that layout already; the one thing the file adds: a symbol written while hunting carries
 *                              the fallback soft decision of its sign dibit (reliability 255, LLR +-255:
 *                              fallback_soft_from_dibit, dsd_dibit.c:592-602, called with soft = NULL from
 *                              frame_sync_capture_symbol, src/dsp/dsd_frame_sync.c:2130-2149), where the loops leave zeros.
 *   ddn_wav_write_s16          a RIFF / WAVE PCM16 file (what -w produces through libsndfile in the reference), with the
 *                              float -> int16 rule of the reference's float output path (scale 32767, round to nearest,
 *                              saturate).
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "ddn_hip.h"
#include "ddn_internal.h"

int
ddn_symbol_capture_write(const char* path, const uint8_t* records10, const uint8_t* flags, size_t count, int append) {
    if (!path || (count && (!records10 || !flags))) {
        return DDN_EINVAL;
    }
    FILE* f = fopen(path, append ? "ab" : "wb");
    if (!f) {
        ddn_set_error("ddn_symbol_capture_write: cannot open '%s'", path);
        return DDN_EINVAL;
    }
    int rc = DDN_OK;
    if (!append) {
        const unsigned char header[16] = {'D', 'S', 'D', 'N', 'S', 'Y', 'M', '2', 2, 10, 0, 0, 0, 0, 0, 0};
        if (fwrite(header, 1, sizeof(header), f) != sizeof(header)) {
            rc = DDN_EHIP;
        }
    }
    for (size_t i = 0; i < count && rc == DDN_OK; i++) {
        unsigned char r[10];
        memcpy(r, records10 + i * 10, 10);
        if (!(flags[i] & 1)) { /* hunting symbol: fallback_soft_from_dibit(dibit, 255) */
            const int d = r[0] & 3;
            const int l0 = ((d >> 1) & 1) ? 255 : -255, l1 = (d & 1) ? 255 : -255;
            r[1] = 255;
            r[2] = (unsigned char)((unsigned)l0 & 0xFFu);
            r[3] = (unsigned char)(((unsigned)l0 >> 8) & 0xFFu);
            r[4] = (unsigned char)((unsigned)l1 & 0xFFu);
            r[5] = (unsigned char)(((unsigned)l1 >> 8) & 0xFFu);
        }
        if (fwrite(r, 1, 10, f) != 10) {
            rc = DDN_EHIP;
        }
    }
    if (fclose(f) != 0 && rc == DDN_OK) {
        rc = DDN_EHIP;
    }
    if (rc != DDN_OK) {
        ddn_set_error("ddn_symbol_capture_write: write to '%s' failed", path);
    }
    return rc;
}

static void
put_le(unsigned char* p, unsigned long v, int n) {
    for (int i = 0; i < n; i++) {
        p[i] = (unsigned char)((v >> (8 * i)) & 0xFFu);
    }
}

int
ddn_wav_write_s16(const char* path, int sample_rate_hz, int channels, const float* pcm, size_t frames, float full_scale) {
    if (!path || !pcm || sample_rate_hz <= 0 || channels <= 0 || channels > 2 || !(full_scale > 0.0f)) {
        return DDN_EINVAL;
    }
    const size_t n = frames * (size_t)channels;
    if (n > 0x7FFFFFF0u / 2) {
        return DDN_ERANGE;
    }
    FILE* f = fopen(path, "wb");
    if (!f) {
        ddn_set_error("ddn_wav_write_s16: cannot open '%s'", path);
        return DDN_EINVAL;
    }
    unsigned char h[44];
    memcpy(h, "RIFF", 4);
    put_le(h + 4, 36 + n * 2, 4);
    memcpy(h + 8, "WAVEfmt ", 8);
    put_le(h + 16, 16, 4);
    put_le(h + 20, 1, 2); /* PCM */
    put_le(h + 22, (unsigned long)channels, 2);
    put_le(h + 24, (unsigned long)sample_rate_hz, 4);
    put_le(h + 28, (unsigned long)sample_rate_hz * (unsigned long)channels * 2, 4);
    put_le(h + 32, (unsigned long)channels * 2, 2);
    put_le(h + 34, 16, 2);
    memcpy(h + 36, "data", 4);
    put_le(h + 40, n * 2, 4);
    int rc = fwrite(h, 1, 44, f) == 44 ? DDN_OK : DDN_EHIP;
    for (size_t i = 0; i < n && rc == DDN_OK; i++) {
        float v = pcm[i] / full_scale * 32767.0f;
        long q = lrintf(v);
        q = q > 32767 ? 32767 : (q < -32768 ? -32768 : q);
        unsigned char b[2];
        put_le(b, (unsigned long)(q & 0xFFFF), 2);
        if (fwrite(b, 1, 2, f) != 2) {
            rc = DDN_EHIP;
        }
    }
    if (fclose(f) != 0 && rc == DDN_OK) {
        rc = DDN_EHIP;
    }
    return rc;
}

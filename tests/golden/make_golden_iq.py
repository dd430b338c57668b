"""Copies two of the reference's own I/Q regression captures (data files under tests/fixtures/iq, cu8 @48 ksps) into
tests/golden together with the known answers the reference's full-chain tests assert on them
(tests/CMakeLists.txt:8888-8900: DECODE_IQ_P25P1_C4FM_CC expects "NAC/CC: 140", DECODE_IQ_P25P1_C4FM_VOICE expects
"Group Voice Channel User", i.e. a voice call: LDU1/LDU2 frames).  Run in the build container only."""
import os

import numpy as np

SRC = "/root/reference/tests/fixtures/iq"
HERE = os.path.dirname(os.path.abspath(__file__))
for name, nac_hex, note in (("p25p1_c4fm_cc", "140", "control channel: TSDU frames, expected payload field NAC/CC: 140"),
                            ("p25p1_c4fm_vc", "", "voice channel: LDU1/LDU2 frames (Group Voice Channel User)"),
                            ("p25p1_cqpsk_cc", "", "CQPSK/LSM control channel; DECODE_IQ_P25P1_CQPSK_CC expects "
                                                   "'WACN: 92065; SYS: 0D5' (tests/CMakeLists.txt:8901-8906)"),
                            ("p25p1_cqpsk_vc", "", "CQPSK/LSM voice channel; DECODE_IQ_P25P1_CQPSK_VOICE expects 'Group Voice "
                                                   "Channel User' (tests/CMakeLists.txt:8907-8912)"),
                            ("p25p1_cqpsk_cc_simulcast", "", "two-ray simulcast impairment of the CQPSK control channel; "
                                                             "DECODE_IQ_P25P1_CQPSK_SIMULCAST_CC expects 'Group Voice Channel Grant "
                                                             "Update - Implicit' (tests/CMakeLists.txt:8913-8921)"),
                            ("dmr_voice", "", "DMR BS voice call; DECODE_IQ_DMR_VOICE expects 'Color Code=02' (tests/CMakeLists.txt:8925-8929)"),
                            ("dmr_t3_cc", "", "DMR Tier III control channel (CSBK data bursts); DECODE_IQ_DMR_T3_CC expects "
                                              "'Color Code=02' (tests/CMakeLists.txt:8930)"),
                            ("dmr_t3_ras_cc", "", "CSBK-only RAS TSCC, colour code 0; DECODE_IQ_DMR_T3_RAS_CC[_COLOR_CODE] expect "
                                                  "'C_ALOHA_SYS_PARMS: Large; Net ID: 1; Site ID: 1' and 'Color Code=00' (tests/CMakeLists.txt:8936-8947)"),
                            ("nxdn48", "", "NXDN48 (2400 baud); DECODE_IQ_NXDN48 expects 'Src=901' (tests/CMakeLists.txt:8948)"),
                            ("nxdn96", "", "NXDN96 (4800 baud); DECODE_IQ_NXDN96 (-fn) expects 'RAN 00' (tests/CMakeLists.txt:8949)"),
                            ("m17", "", "M17 stream transmission (preamble, LSF, stream frames, EOT); DECODE_IQ_M17 (-fz) expects "
                                        "'SRC: N0CALL' (tests/CMakeLists.txt:8964)"),
                            ("ysf", "", "Yaesu System Fusion, V/D mode 2 communication channel frames; DECODE_IQ_YSF (-fy) expects "
                                        "'V/D2 RID Mode Repeater CC' (tests/CMakeLists.txt:8953-8957)"),
                            ("p25p2_cc", "", "P25 Phase 2 TDMA control channel; DECODE_IQ_P25P2_CC (-f2) expects 'P25p2 SACCH' "
                                             "(tests/CMakeLists.txt:8923)")):
    iq = np.fromfile(os.path.join(SRC, name + ".iq"), dtype=np.uint8).reshape(-1, 2)
    np.savez_compressed(os.path.join(HERE, "iq_%s.npz" % name), iq=iq, rate=np.int32(48000),
                        expected_nac_hex=np.bytes_(nac_hex), note=np.bytes_(note))
    print(name, iq.shape)

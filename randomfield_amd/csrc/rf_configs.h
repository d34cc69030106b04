// rf_configs.h -- which radices / tile shapes each supported axis length uses.
// Shared by the HIP dispatch (rf_kernels.hip) and the CPU emulator.
#pragma once
#include "rf_fft.h"

namespace rf {

// ---- strided (x / y) passes: N = axis length -------------------------------
// TC is the tile width in columns for float32 (float64 uses TC/2 so that the
// LDS footprint and the 16-byte-per-lane access width stay the same).
template <typename T, int N> struct ColSel;
#define RF_COL(N, R1, R2, R3, TC32, NT)                                                         \
  template <> struct ColSel<float, N>  { using type = ColCfg<float,  N, R1, R2, R3, TC32,     NT>; }; \
  template <> struct ColSel<double, N> { using type = ColCfg<double, N, R1, R2, R3, TC32 / 2, NT>; };
RF_COL(8,    8,  1,  1, 32, 256)
RF_COL(16,   16, 1,  1, 32, 256)
RF_COL(32,   8,  4,  1, 32, 256)
RF_COL(64,   8,  8,  1, 32, 256)
RF_COL(128,  16, 8,  1, 32, 256)
RF_COL(256,  16, 16, 1, 16, 256)
RF_COL(512,  8,  8,  8, 16, 512)
// float64, N = 1024: 8-column tiles (whole 128-byte lines, one 1024-thread workgroup per CU) instead of the TC32 / 2 rule's
// 4 columns -- measured on MI355X (tools/f64_prof.py): generation pass 4.86 -> 3.25 ms, y pass 3.5 -> 3.3 ms per 1024^3
#ifndef RF_COL64_1024
#define RF_COL64_1024 16, 8, 8, 8, 1024
#endif
#ifndef RF_COL32_1024
#define RF_COL32_1024 16, 8, 8, 8, 512
#endif
template <> struct ColSel<float, 1024>  { using type = ColCfg<float,  1024, RF_COL32_1024>; };
template <> struct ColSel<double, 1024> { using type = ColCfg<double, 1024, RF_COL64_1024>; };
#ifndef RF_COL32_2048
#define RF_COL32_2048 8, 16, 16, 8, 1024
#endif
#ifndef RF_GEN32_2048
#define RF_GEN32_2048 RF_COL32_2048
#endif
template <> struct ColSel<float, 2048>  { using type = ColCfg<float,  2048, RF_COL32_2048>; };
template <> struct ColSel<double, 2048> { using type = ColCfg<double, 2048, 8, 16, 16, 4, 1024>; };
#undef RF_COL
// fused-generation x pass: same tiles, radices chosen for register pressure (generation happens
// in pass 1, so a small first radix keeps the live set low)
template <typename T, int N> struct GenSel { using type = typename ColSel<T, N>::type; };
template <> struct GenSel<float, 1024> { using type = ColCfg<float, 1024, 8, 16, 8, 8, 512>; };
template <> struct GenSel<float, 2048> { using type = ColCfg<float, 2048, RF_GEN32_2048>; };
#ifndef RF_GEN64_1024
#define RF_GEN64_1024 8, 8, 16, 8, 1024
#endif
template <> struct GenSel<double, 1024> { using type = ColCfg<double, 1024, RF_GEN64_1024>; };
// the half-length configuration of the float32 in-place pass of length 1024 (rf_k_col_plain.hip RF_Y_COL2_1024; the emulator follows)
struct PairSel1024 { using type = ColCfg<float, 512, 8, 8, 8, 8, 256>; };
#ifndef RF_Y_COL2_1024
#define RF_Y_COL2_1024 1               // (rf_k_col_plain.hip launch_col_plain, rf_k_yz.hip: the in-place float32 pass of length 1024 in that form)
#endif
#ifndef RF_COL2_2048
#define RF_COL2_2048 1                 // length-2048 float32 in-place / direct passes as two 1024-point transforms per tile (Col2); 0 = the whole-column kernels
#endif
#define RF_COL_SIZES(X) X(8) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048)

// ---- contiguous (z) pass: M = nz / 2 ----------------------------------------
template <typename T, int M> struct RowSel;
// float32 and float64 get their own radices: a float64 butterfly pair of radix 16 needs 256+ VGPRs
#define RF_ROW(M, R1, R2, R3, NRT32, D1, D2, D3, NRT64, NT)                                        \
  template <> struct RowSel<float, M>  { using type = RowCfg<float,  M, R1, R2, R3, NRT32, NT>; }; \
  template <> struct RowSel<double, M> { using type = RowCfg<double, M, D1, D2, D3, NRT64, NT>; };
RF_ROW(8,    8,  1,  1,  256, 8,  1,  1,  128, 256)
RF_ROW(16,   16, 1,  1,  256, 8,  2,  1,  64,  256)
RF_ROW(32,   8,  4,  1,  64,  8,  4,  1,  32,  256)
RF_ROW(64,   8,  8,  1,  64,  8,  8,  1,  32,  256)
RF_ROW(128,  8,  16, 1,  32,  8,  4,  4,  16,  256)
RF_ROW(256,  8,  8,  4,  16,  8,  8,  4,  8,   256)
#ifndef RF_ROW64_512
#define RF_ROW64_512 8, 8, 8, 4, 256
#endif
#ifndef RF_ROW32_512
#define RF_ROW32_512 8, 8, 8, 8, 256
#endif
template <> struct RowSel<float, 512>  { using type = RowCfg<float,  512, RF_ROW32_512>; };
template <> struct RowSel<double, 512> { using type = RowCfg<double, 512, RF_ROW64_512>; };
RF_ROW(1024, 8,  16, 8,  4,   8,  8,  16, 2,   256)
// rows of 2048 complex: only the unpacked c2c transform has them (a packed plan's rows hold nz/2 <= 1024)
RF_ROW(2048, 8,  16, 16, 2,   8,  16, 16, 1,   256)
#undef RF_ROW
#define RF_ROW_SIZES(X) X(8) X(16) X(32) X(64) X(128) X(256) X(512) X(1024)
#define RF_ROWC_SIZES(X) RF_ROW_SIZES(X) X(2048)

inline bool col_size_supported(int n) {
  switch (n) {
#define X(N) case N:
    RF_COL_SIZES(X)
#undef X
    return true;
    default: return false;
  }
}
inline bool row_size_supported(int m) {
  switch (m) {
#define X(M) case M:
    RF_ROW_SIZES(X)
#undef X
    return true;
    default: return false;
  }
}
inline bool rowc_size_supported(int m) { return m == 2048 || row_size_supported(m); }   // unpacked c2c rows

}  // namespace rf

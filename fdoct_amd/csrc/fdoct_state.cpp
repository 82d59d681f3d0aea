// fdoct_state.cpp -- plan selection and the device state of a handle: which kernel family a configuration takes
// (select_plan / select_generic), and everything the kernels read that is built on the host in double and uploaded once per
// change of the handle's state (reciprocal words of the background and their half-float pattern, window and slope planes,
// gather tables, twiddles, Bluestein chirps).  Part of the C-ABI layer (fdoct_ctx.h); no CPU compute path.
#include "fdoct_ctx.h"

namespace fdoct_impl {

size_t dtype_size(int dt) {
  switch (dt) {
    case FDOCT_U8: return 1;
    case FDOCT_U16: return 2;
    case FDOCT_F32: return 4;
    case FDOCT_F64: return 8;
    default: return 0;
  }
}

int copy_ref_frame(fdoct_ctx* h, RefFrame& dst, const void* data, fdoct_dtype dtype, int rows, size_t pitch) {
  if (!data) {
    dst.v.clear();
    dst.rows = 0;
    h->dirty = true;
    return FDOCT_OK;
  }
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  if (rows != 1 && rows != h->H) return fail(h, FDOCT_ERR_INVALID, "reference frame rows must be 1 or height");
  if (pitch == 0) pitch = es * h->W;
  if (pitch < es * h->W) return fail(h, FDOCT_ERR_INVALID, "pitch smaller than a row");
  dst.v.resize((size_t)rows * h->W);
  for (int r = 0; r < rows; r++) {
    const unsigned char* row = static_cast<const unsigned char*>(data) + (size_t)r * pitch;
    double* o = dst.v.data() + (size_t)r * h->W;
    for (int i = 0; i < h->W; i++) {
      switch (dtype) {
        case FDOCT_U8: o[i] = reinterpret_cast<const uint8_t*>(row)[i]; break;
        case FDOCT_U16: o[i] = reinterpret_cast<const uint16_t*>(row)[i]; break;
        case FDOCT_F32: o[i] = reinterpret_cast<const float*>(row)[i]; break;
        default: o[i] = reinterpret_cast<const double*>(row)[i]; break;
      }
    }
  }
  dst.rows = rows;
  h->dirty = true;
  return FDOCT_OK;
}

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// n = 2^a 3^b 5^c -> Stockham radices (4s first), false if another prime divides n
// Radix plan of the generic kernel's Stockham DFT (radices 16/8/4/2/5/3).  The first pass writes butterfly j's
// outputs R apart (stride R*8 bytes across lanes), so it gets an odd radix -- or a small power of two -- to keep
// those LDS writes off the same banks; it is also the pass without twiddle multiplies.
bool factor_radices(int n, std::vector<int>& rad, int log2max) {
  rad.clear();
  int a = 0, b = 0, c = 0;
  while (n % 2 == 0) { a++; n /= 2; }
  while (n % 3 == 0) { b++; n /= 3; }
  while (n % 5 == 0) { c++; n /= 5; }
  if (n != 1) return false;
  for (int i = 0; i < c; i++) rad.push_back(5);
  for (int i = 0; i < b; i++) rad.push_back(3);
  if (rad.empty() && a > 0) {
    const int first = (a % 2) ? 1 : 2;
    rad.push_back(1 << first);
    a -= first;
  }
  const int kLog2Max = log2max ? log2max : (GENERIC_MAX_RADIX >= 16 ? 4 : 3);
  for (; a >= kLog2Max; a -= kLog2Max) rad.push_back(1 << kLog2Max);
  if (a) rad.push_back(1 << a);
  return (int)rad.size() <= GENERIC_MAX_PASSES;
}

// real rows run the N-point DFT as an N/2-point complex one (see generic_kernel)
bool generic_real_half(const fdoct_ctx* h) { return h->phase.empty() && (h->N % 2) == 0; }

int generic_buffer_len(const fdoct_ctx* h) {
  const int MW = h->W * h->M;
  int L = generic_real_half(h) ? h->N / 2 : h->N;
  if (h->M > 1) L = std::max(L, MW / 2);  // the zero-pad DFTs run at half length (real row, Hermitian spectrum)
  if (h->blu_m > L) L = h->blu_m;         // Bluestein: the transform runs as two power-of-two DFTs of this length
  if (h->M > 1 && h->zp_full) {           // the zero-pad stage at full length: both transforms, and the upsampled row as floats
    L = std::max(L, std::max(h->gzf.blu_m ? h->gzf.blu_m : h->W, h->gzi.blu_m ? h->gzi.blu_m : h->zn));
    L = std::max(L, (MW + 1) / 2);
  }
  return L;
}

size_t generic_lds_bytes(const fdoct_ctx* h, int buffers) {
  const int L = generic_buffer_len(h);
  const int ybuf = (h->W + 3) & ~3;
  if (!buffers) buffers = h->generic_inplace ? 1 : 2;
  return (size_t)ybuf * 4 + (size_t)L * 8 * buffers + (size_t)((h->D + 3) & ~3) * 4;  // row, the DFT buffer(s), magnitude sums
}

// The any-configuration path: checks that fdoct_generic.hip can run this geometry.
int select_generic(fdoct_ctx* h) {
  const int MW = h->W * h->M;
  // cv::dft takes any length (main:1185).  Lengths with prime factors up to 5 run as mixed-radix Stockham passes; any
  // other length as Bluestein's algorithm: two power-of-two DFTs of length >= 2n - 1 around a chirp multiplication.
  h->blu_m = 0;
  const int tlen = generic_real_half(h) ? h->N / 2 : h->N;  // the transform the kernel actually runs
  std::vector<int> probe;
  if (!factor_radices(tlen, probe)) {
    int mb = 1;
    while (mb < 2 * tlen - 1) mb <<= 1;
    // (round 6: the smallest length 2^a 3^b 5^c >= 2n - 1 where the host can afford the transformed chirp by the DFT's definition:
    // 1296 instead of 2048 around a 642-point transform)
    if (2 * tlen - 1 <= 8192) {
      for (int c = 2 * tlen - 1; c < mb; c++) {
        int m = c;
        for (int f : {2, 3, 5})
          while (m % f == 0) m /= f;
        std::vector<int> tmp;
        if (m == 1 && factor_radices(c, tmp)) {
          mb = c;
          break;
        }
      }
    }
    h->blu_m = mb;
    factor_radices(mb, h->rad_blu);
    h->rad_n.clear();
    h->rad_nh.clear();
  } else {
    if (!factor_radices(h->N, h->rad_n)) h->rad_n.clear();  // (only used when the full-length transform runs)
    if ((h->N % 2) == 0 && !factor_radices(h->N / 2, h->rad_nh)) h->rad_nh.clear();
  }
  h->use_big = false;
  if (h->M > 1) {
    // an odd width (the reference's fftshift leaves the last column of the spectrum where it is and, under an even multiplier,
    // pads to M W - 1 bins, main:215-241) and zero-pad lengths with a prime factor above 5: the long-row path, whose DFTs run at
    // full length and take any length (the LDS kernels halve the transforms of a real row, which needs an even width)
    h->zp_full = false;
    if ((h->W % 2) || !factor_radices(h->W / 2, h->rad_wh) || !factor_radices(MW / 2, h->rad_mwh)) {
      h->rad_wh.clear();
      h->rad_mwh.clear();
      // Round 6: such a row stays in LDS when two buffers of its full-length transforms fit -- the W-point and the padded
      // spectrum's zn-point +i transforms inside generic_kernel (Stockham passes, or Bluestein around a power of two: 321 x 4 ->
      // 1283 points, a prime, runs around 4096) -- and leaves for HBM only when they do not.
      h->zn = h->W + 2 * ((MW - h->W) / 2);
      auto plan = [](int n, fdoct_ctx::GenericDftPlan& p) {
        p.n = n;
        p.blu_m = 0;
        if (!factor_radices(n, p.rad)) {
          // Bluestein around the smallest length 2^a 3^b 5^c >= 2n - 1 (the convolution only needs that much room; the passes
          // take radices 2, 3, 4, 5, 8): 2592 for a 1283-point transform where the next power of two is 4096
          int mb = 2 * n - 1;
          for (;; mb++) {
            int m = mb;
            for (int f : {2, 3, 5})
              while (m % f == 0) m /= f;
            if (m == 1 && factor_radices(mb, p.rad)) break;
            if (mb > 4 * n) return false;
          }
          p.blu_m = mb;
          return mb <= 8192;   // (the host builds the transformed chirp by the DFT's definition: bounded work)
        }
        return true;
      };
      h->zp_full = plan(h->W, h->gzf) && plan(h->zn, h->gzi);
      if (h->zp_full && generic_lds_bytes(h, 2) + 1024 > 160 * 1024) h->zp_full = false;
      static const bool no_full = [] { const char* e = std::getenv("FDOCT_NO_ZP_FULL"); return e && std::atoi(e) != 0; }();  // measurement: round 5's route
      if (no_full) h->zp_full = false;
      if (!h->zp_full) h->use_big = true;
    }
  } else {
    h->zp_full = false;
  }
  // rows whose two DFT buffers do not fit the 160 KB of LDS (half-length transforms beyond about 9000 points): with ONE buffer and
  // every step in place (generic_kernel<1024, 1, true>) up to 16384 points -- 4096 samples upsampled x8 -- as long as a thread of
  // the 1024 holds its share of a pass in 16 registers (radices 5 / 3: 15), the zero-pad spectrum in 8 and the resampled row
  // in 32, and the length needs no Bluestein; what lies beyond runs with the rows in HBM (fdoct_big.hip)
  h->generic_inplace = false;
  // (FDOCT_GENERIC_INPLACE_ABOVE: the two-buffer footprint above which the one-buffer kernel is taken, for measurements)
  static const size_t inplace_above = [] { const char* e = std::getenv("FDOCT_GENERIC_INPLACE_ABOVE"); return e ? (size_t)std::atol(e) : (size_t)160 * 1024; }();
  const bool must_inplace = generic_lds_bytes(h, 2) + 1024 > 160 * 1024;
  if (!h->zp_full && generic_lds_bytes(h, 2) + 1024 > inplace_above) {
    auto pass_ok = [](const std::vector<int>& rad, int n) {
      for (int R : rad)
        if (R > 16 || n / R > 1024 * (16 / R)) return false;
      return !rad.empty();
    };
    const bool real_half = generic_real_half(h);
    // (the in-place passes take radix-16 butterflies -- one per thread on a 16384-point transform -- and with them a pass less)
    std::vector<int> r_n = h->rad_n, r_nh = h->rad_nh, r_wh = h->rad_wh, r_mwh = h->rad_mwh;
    if (!h->blu_m && !h->use_big) {
      if (!h->rad_n.empty()) factor_radices(h->N, h->rad_n, 4);
      if (!h->rad_nh.empty()) factor_radices(h->N / 2, h->rad_nh, 4);
      if (h->M > 1) {
        factor_radices(h->W / 2, h->rad_wh, 4);
        factor_radices(MW / 2, h->rad_mwh, 4);
      }
    }
    const bool ok = !h->use_big && !h->blu_m && generic_lds_bytes(h, 1) + 1024 <= 160 * 1024 && h->N <= 32 * 1024 &&
                    (real_half ? pass_ok(h->rad_nh, h->N / 2) : pass_ok(h->rad_n, h->N)) &&
                    (h->M == 1 || (h->W / 2 <= 8 * 1024 && pass_ok(h->rad_wh, h->W / 2) && pass_ok(h->rad_mwh, MW / 2)));
    if (ok) {
      h->generic_inplace = true;
    } else {
      if (must_inplace) h->use_big = true;
      h->rad_n = r_n; h->rad_nh = r_nh; h->rad_wh = r_wh; h->rad_mwh = r_mwh;
    }
  }
  // rows of which a CU holds one (two buffers beyond half the LDS) run with 1024 threads, 128 registers each: radix-16 passes there too
  h->generic_radix16 = h->generic_inplace;
  {
    static const int r16 = [] { const char* e = std::getenv("FDOCT_GENERIC_RADIX16"); return e ? std::atoi(e) : 1; }();  // measurement
    if (r16 && !h->generic_inplace && !h->use_big && !h->blu_m && !h->zp_full && generic_lds_bytes(h, 2) > (160 * 1024 - 1024) / 2) {
      if (!h->rad_n.empty()) factor_radices(h->N, h->rad_n, 4);
      if (!h->rad_nh.empty()) factor_radices(h->N / 2, h->rad_nh, 4);
      if (h->M > 1) {
        factor_radices(h->W / 2, h->rad_wh, 4);
        factor_radices(MW / 2, h->rad_mwh, 4);
      }
      h->generic_radix16 = true;
    }
  }
  {
    static const int force = [] { const char* e = std::getenv("FDOCT_FORCE_LONG_ROWS"); return e ? std::atoi(e) : 0; }();  // measurement
    if (force || h->plan_override == -3) h->use_big = true, h->generic_inplace = false, h->zp_full = false;   // (-3: fdoct_set_plan's "rows in HBM")
  }
  if (h->use_big && (h->N > (1 << 24) || MW > (1 << 24)))
    return fail(h, FDOCT_ERR_UNSUPPORTED, "rows of more than 2^24 points");
  h->use_generic = true;
  return FDOCT_OK;
}

// Pick the compiled plan for the current (N, W, phase) and derive LDS geometry; configurations without a
// specialised kernel go to the generic path.
int select_plan(fdoct_ctx* h) {
  h->cplx = !h->phase.empty();
  h->use_generic = false;
  h->NC = h->cplx ? h->N : h->N / 2;
  const bool special_ok = is_pow2(h->N) && h->M == 1 && (h->W % 8) == 0 && (h->cplx || h->D <= h->N / 2) &&
                          h->plan_override > -2;
  bool found = false;
  // preference order for equal NC: the override, then the measured-fastest plan ids
  static const int pref[] = {5, 2, 3, 0, 1, 7, 6, 8, 4};  // per NC: fastest first; equal plans: smallest chunk count that holds W
  FusedPlan q{};
  if (special_ok && h->plan_override >= 0 && fused_plan_get(h->plan_override, &q) && q.nc == h->NC && h->W <= 8 * q.T * q.WCH) {
    h->plan = q;
    found = true;
  }
  for (int i = 0; special_ok && !found && i < (int)(sizeof pref / sizeof pref[0]); i++) {
    if (fused_plan_get(pref[i], &q) && q.nc == h->NC && h->W <= 8 * q.T * q.WCH) {
      h->plan = q;
      found = true;
    }
  }
  if (!found) return select_generic(h);
  const FusedPlan& p = h->plan;
  const int WC = 8 * p.T * p.WCH;
  const int LP = p.R1 == 32 ? 5 : p.R1 == 16 ? 4 : p.R1 == 8 ? 3 : 2;
  const int stg = 4 * (WC + 4);
  const int xch = p.kind == 1 ? 8 * (65 * 16 + 2) : p.kind == 2 ? 8 * (129 * 16 + 2) : 8 * (h->NC + (h->NC >> LP) + 2);
  h->scratch_bytes = ((stg > xch ? stg : xch) + 15) & ~15;
  const double sigma = (h->cplx ? 1.0 : 2.0) * (double)(h->W * h->M) / (double)h->N;
  h->split = (sigma >= 1.5 && sigma <= 3.0) ? 1 : 0;
  int tw = (p.R2 - 1) * p.R1 + (p.R3 > 1 ? (p.R3 - 1) * p.R1 * p.R2 : 0);
  if (p.kind == 1) tw = 48 + 15 * 64;
  if (p.kind == 2) tw = 96 + 128;  // step-5 twiddles are formed as powers of W_2048^(l') in the kernel
  h->tw_count = (tw + 1) & ~1;
  return FDOCT_OK;
}

// planes: the three constant planes are staged in LDS (kernels that do not keep them in registers); il_plane: so is the low
// word of the reciprocal background (FusedArgs::prec == 1)
// il_half: that plane holds half floats (the fast-path kernels with at most 32 samples per lane: fused_il_half)
// tw3 / gi: the step-5 twiddle table and the gather table are staged (the transposed-store kernels leave out what they hold in
// registers: fused_tw3_in_lds / fused_gi_in_lds)
size_t const_lds_bytes(const fdoct_ctx* h, bool planes, bool il_plane, bool il_half, bool tw3, bool gi) {
  const int WC = 8 * h->plan.T * h->plan.WCH;
  const size_t tw_entries = tw3 ? (size_t)h->tw_count : (size_t)(h->plan.R2 - 1) * h->plan.R1;
  return (planes ? (size_t)3 : 0) * WC * 4 + (il_plane ? (size_t)WC * (il_half ? 2 : 4) : 0) + tw_entries * 8 + (h->cplx ? (size_t)h->NC * 8 : 0) + (gi ? (size_t)h->NC * 4 : 0);
}
// constants of a transposed-store launch (fast path, 1024-point row-swap plan)
size_t tro_const_lds_bytes(const fdoct_ctx* h, int sample_bytes, bool normalize) {
  const FusedPlan& p = h->plan;
  const bool both = h->precise_div, ib2d = h->yb.rows > 1, half = fused_il_half(true, p.WCH);
  // (the row-swap plan's transposed-store kernels hold the constant planes in registers; the 512-point Stockham plan's read them
  // from LDS like its row-major kernels, and its averaging kernels take the low words from global memory: fused_il_global)
  const bool planes = !fused_resident_consts(p.kind, true, h->A > 1, p.WCH, 0);
  const bool il_plane = both && !ib2d && !fused_il_global(true, h->A > 1, p.WCH, p.T);
  return const_lds_bytes(h, planes, il_plane, half, fused_tw3_in_lds(p.kind, true, 0, true, ib2d && both && half),
                         fused_gi_in_lds(p.kind, true, 0, false, h->A > 1, true,
                                         fused_tro_pf2(p.kind, true, 0, false, h->A > 1, true, ib2d, normalize ? 1 : 0, sample_bytes)));
}

// main:1132 divides by data_yb in double.  The kernels multiply by the reciprocal, held as an unevaluated sum of two floats
// ib + il = 1/yb to 2^-48: ib = fl32(1/yb) alone is off by up to 6e-8 of the quotient -- a fixed per-column pattern of the
// size of the DC level, which the chain turns into up to 4e-6 of the DC level per depth bin: more than the whole tolerance
// once the fringes are weaker than about 1 % of it.  With d = fma(v, ib, -c0) (rounded at the size of the deviation from the
// mean estimate c0) followed by d = fma(v, il, d), nothing is rounded at the size of the DC level.  x/0 -> 0 (OpenCV 3.x
// Mat division).
// float -> IEEE half bits, round to nearest even (values here are at most 2^14 in magnitude: no overflow handling needed beyond inf)
uint16_t half_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t e = (int32_t)((x >> 23) & 0xffu) - 127 + 15;
  uint32_t m = x & 0x7fffffu;
  if (((x >> 23) & 0xffu) == 0xffu) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {  // subnormal half (or zero)
    if (e < -10) return (uint16_t)sign;
    m |= 0x800000u;
    const int shift = 14 - e;  // 13 + (1 - e)
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;  // (a carry into the exponent is the right result)
  return (uint16_t)(sign | r);
}

// The second word as the fast-path kernels with at most 32 samples per lane apply it (fdoct_kernels.h: FDOCT_PREC16): what
// v * ib leaves out of v / yb is (v * ib) * rho, rho = (1/yb - ib) / ib, |rho| <= 2^-24; the kernel adds c0 * rho (c0: its
// estimate of the row mean of v / yb).  rho * 2^38 as half floats, in the order the lanes read them: the lane's 8-sample group
// of chunk c is 16 bytes at ((c T + lane) * 16), dword q = samples chunk_pair_offset(q), + 2 (the RawChunk pair order).
void half_pattern_row(const double* yb, int WC, int T, uint32_t* out) {
  auto rho_h = [&](int i) -> uint16_t {
    if (yb[i] == 0.0) return 0;
    const double q = 1.0 / yb[i];
    const float ib = (float)q;
    if (!std::isfinite(ib) || ib == 0.f) return 0;
    return half_bits((float)(std::ldexp((q - (double)ib) / (double)ib, kPrec16Shift)));
  };
  for (int i0 = 0; i0 < WC; i0 += 8) {
    const int grp = i0 / 8, ln = grp % T, c = grp / T;
    for (int q = 0; q < 4; q++) {
      const int off = (q & 1) * 4 + (q >> 1);
      out[(size_t)(c * T + ln) * 4 + q] = (uint32_t)rho_h(i0 + off) | ((uint32_t)rho_h(i0 + off + 2) << 16);
    }
  }
}

void reciprocal_words(const std::vector<double>& yb, std::vector<float>& ib, std::vector<float>& il) {
  ib.resize(yb.size());
  il.resize(yb.size());
  for (size_t i = 0; i < yb.size(); i++) {
    if (yb[i] != 0.0) {
      const double q = 1.0 / yb[i];
      ib[i] = (float)q;
      const double lo = q - (double)ib[i];
      il[i] = std::isfinite(lo) ? (float)lo : 0.f;  // (1/yb beyond the float range: ib is inf, as before)
    } else {
      ib[i] = il[i] = 0.f;
    }
  }
}

// smoothmovavg (main:247-304, 990-991) divides its 2n + 2 taps by 2 (n + 1) in double.  An f32 quotient would be a rounding at the
// size of the DC level unless n + 1 is a power of two (5 x the tolerance on fringes of 0.1 % of it with n = 2), so the pass in front
// of the chain hands on the tap SUMS -- exact in f32 for the camera's integer samples up to n = 126 -- and the factor K = 2 (n + 1)
// goes where the reference's arithmetic puts it: into the dark frame (subtracted from the samples themselves), and, unless a min-max
// normalisation follows (it is scale-invariant), into the pi frame and the background as well.
PlaneScales plane_scales(const fdoct_ctx* h) {
  const double K = h->cfg.movavgn > 0 ? 2.0 * ((double)h->cfg.movavgn + 1.0) : 1.0;
  const bool norm_on = h->cfg.rowwisenormalize || (h->cfg.variant == FDOCT_VARIANT_SIM) || !h->cfg.donotnormalize;
  return {norm_on ? 1.0 : K, norm_on ? 1.0 : K, K};
}
std::vector<double> scaled_copy(const std::vector<double>& v, double s) {
  std::vector<double> t(v);
  if (s != 1.0)
    for (double& x : t) x *= s;
  return t;
}

// Recompute everything the kernel reads from the host-side state and upload it.
int rebuild_generic_state(fdoct_ctx* h);

int rebuild_device_state(fdoct_ctx* h) {
  int rc = select_plan(h);
  if (rc) return rc;
  h->generic_tables_ok = false;
  h->wave_tables_ok = false;
  if (h->use_generic) return rebuild_generic_state(h);
  const int W = h->W, H = h->H, N = h->N;
  const FusedPlan& p = h->plan;
  const int WC = 8 * p.T * p.WCH;
  DEVICE_SCOPE(h);

  // 1/background in double, as two floats (reciprocal_words)
  {
    std::vector<float> ib, il;
    const std::vector<double> ybs = h->yb.rows ? scaled_copy(h->yb.v, plane_scales(h).yb) : std::vector<double>();
    if (h->yb.rows) reciprocal_words(ybs, ib, il);
    {  // the half-float pattern of the second word (rows exactly one chunk width wide: the fast path's condition)
      std::vector<uint32_t> h16, h16_2d;
      if (W == WC && h->yb.rows == 1) {
        h16.resize((size_t)WC / 2);
        half_pattern_row(ybs.data(), WC, p.T, h16.data());
      } else if (W == WC && h->yb.rows > 1) {
        h16_2d.resize((size_t)H * WC / 2);
        for (int r = 0; r < H; r++) half_pattern_row(ybs.data() + (size_t)r * W, WC, p.T, h16_2d.data() + (size_t)r * WC / 2);
      }
      if ((rc = upload(h, &h->d_il16, h16))) return rc;
      if ((rc = upload(h, &h->d_il16_2d, h16_2d))) return rc;
    }
    if (h->yb.rows == 1) {
      if ((rc = upload(h, &h->d_ib, ib))) return rc;
      if ((rc = upload(h, &h->d_il, il))) return rc;
      {  // the same plane in the slot order of the kernels' LDS planes (sample 8 (ln + T c) + e -> c 8T + (e & 1) 4T + 4 ln + (e >> 1))
        std::vector<float> ilp((size_t)WC, 0.f);
        for (int i = 0; i < W; i++) {
          const int e = i & 7, ln = (i >> 3) & (p.T - 1), c = i / (8 * p.T);
          ilp[(size_t)c * 8 * p.T + (e & 1) * 4 * p.T + 4 * ln + (e >> 1)] = il[i];
        }
        if ((rc = upload(h, &h->d_il_p, ilp))) return rc;
      }
      if ((rc = dev_alloc(h, &h->d_ib2d_f, 0))) return rc;
      if ((rc = dev_alloc(h, &h->d_il2d_f, 0))) return rc;
    } else {
      // the fused kernels read a 2-D background with every 8-sample group stored evens first, then odds (the
      // order their sample pairs are held in), rows padded to the plan's chunk width; the generic kernel keeps
      // its own natural-order copy (d_ib2d)
      std::vector<float> perm((size_t)H * WC, 0.f);
      auto permute = [&](const std::vector<float>& src) {
        for (int r = 0; r < H; r++)
          for (int i = 0; i < W; i++) perm[(size_t)r * WC + (i & ~7) + ((i & 1) * 4 + ((i & 7) >> 1))] = src[(size_t)r * W + i];
      };
      permute(ib);
      if ((rc = upload(h, &h->d_ib2d_f, perm))) return rc;
      permute(il);
      if ((rc = upload(h, &h->d_il2d_f, perm))) return rc;
      if ((rc = dev_alloc(h, &h->d_ib, 0))) return rc;
      if ((rc = dev_alloc(h, &h->d_il, 0))) return rc;
    }
  }
  // (second words: what the float planes leave of the double ones -- read where the row is formed in double, BscanDark's band-pass)
  auto up_ref = [&](const RefFrame& f, double scale, float** d, float** d_lo) -> int {
    std::vector<float> t(f.v.size()), tl(f.v.size());
    for (size_t i = 0; i < t.size(); i++) {
      t[i] = (float)(f.v[i] * scale);
      tl[i] = (float)(f.v[i] * scale - (double)t[i]);
    }
    if (int e = upload(h, d_lo, tl)) return e;
    return upload(h, d, t);
  };
  if ((rc = up_ref(h->yp, plane_scales(h).yp, &h->d_yp, &h->d_yp_lo))) return rc;
  if ((rc = up_ref(h->yd, plane_scales(h).yd, &h->d_yd, &h->d_yd_lo))) return rc;
  {
    // Window (main:1142) and slope step (main:1153-1173) folded into two per-sample planes: with t = x - mean and
    // y = t * w, s_i = y_i + g_i (y_i - y_(i-1)) = a_i t_i + b_i t_(i-1), a_i = (1 + g_i) w_i, b_i = -g_i w_(i-1).
    // Sample 0 has slopes[0] = slopes[1] (main:1161): s_0 = (1 - g_0) w_0 t_0 + g_0 w_1 t_1; the kernel feeds t_1 there.
    // g_i = fractionalk[i]: the reference indexes fractionalk (N entries) by nearestkindex[q], a SAMPLE index; past N
    // it is out of bounds there and defined as 0 here.  Real path: the 1/2 of the real-input untangle is folded into
    // the window (exact: power of two).  Products in double, rounded once.
    std::vector<float> pa(W), pb(W);
    const double half = h->cplx ? 1.0 : 0.5;
    auto gg = [&](int i) { return i < N ? h->frac[i] : 0.0; };
    for (int i = 1; i < W; i++) {
      pa[i] = (float)((1.0 + gg(i)) * half * h->win[i]);
      pb[i] = (float)(-gg(i) * half * h->win[i - 1]);
    }
    pa[0] = (float)((1.0 - gg(0)) * half * h->win[0]);
    pb[0] = (float)(gg(0) * half * h->win[1]);
    if ((rc = upload(h, &h->d_win, pa))) return rc;
    if ((rc = upload(h, &h->d_g, pb))) return rc;
  }
  {
    // gather sources: data_ylin[q] = s[nearestkindex[q]] for q = 1..N-2, else 0 (main:1164)
    std::vector<uint32_t> gi(h->NC);
    auto off = [&](int q) -> uint32_t {
      if (q <= 0 || q >= N - 1) return (uint32_t)(4 * WC);
      return (uint32_t)staging_offset_bytes(h->idx[q], WC, h->split);
    };
    for (int n = 0; n < h->NC; n++) gi[n] = h->cplx ? off(n) : (off(2 * n) | (off(2 * n + 1) << 16));
    if ((rc = upload(h, &h->d_gidx, gi))) return rc;
  }
  {
    std::vector<float2> tw(h->tw_count, make_float2(0.f, 0.f));
    size_t o = 0;
    if (p.kind == 1 || p.kind == 2) {
      // row-swap plans: tw2[(3c + i-1)*4 + j] = W_(4Q)^(i*(4c+j)), c < Q/4; tw3[(b-1)*L + l] = W_NC^(b*l), l < L = NC/16
      // (Q = first radix: 16 for fft1024_rowswap, 32 for fft2048_rowswap)
      const int Q = p.R1, L = h->NC / 16;
      for (int c = 0; c < Q / 4; c++)
        for (int i = 1; i < 4; i++)
          for (int j = 0; j < 4; j++) {
            const double a = 2.0 * kPi * (double)(i * (4 * c + j)) / (double)(4 * Q);
            tw[(3 * c + i - 1) * 4 + j] = make_float2((float)std::cos(a), (float)std::sin(a));
          }
      for (int b = 1; b < (p.kind == 1 ? 16 : 2); b++)  // kind 2 keeps only the b = 1 row
        for (int l = 0; l < L; l++) {
          const double a = 2.0 * kPi * (double)(b * l) / (double)h->NC;
          tw[3 * Q + (b - 1) * L + l] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    } else
    for (int r = 1; r < p.R2; r++)
      for (int k = 0; k < p.R1; k++) {
        const double a = 2.0 * kPi * (double)r * (double)k / (double)(p.R1 * p.R2);
        tw[o++] = make_float2((float)std::cos(a), (float)std::sin(a));
      }
    if (p.kind == 0 && p.R3 > 1)
      for (int r = 1; r < p.R3; r++)
        for (int k = 0; k < p.R1 * p.R2; k++) {
          const double a = 2.0 * kPi * (double)r * (double)k / (double)h->NC;
          tw[o++] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    if ((rc = upload(h, &h->d_tw, tw))) return rc;
    std::vector<float2> utw(p.T);
    for (int l = 0; l < p.T; l++) {
      const double a = 2.0 * kPi * (double)l / (double)N;
      utw[l] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    if ((rc = upload(h, &h->d_utw, utw))) return rc;
  }
  {
    std::vector<float2> ph(h->phase.size() / 2);
    for (size_t i = 0; i < ph.size(); i++) ph[i] = make_float2(h->phase[2 * i], h->phase[2 * i + 1]);
    if ((rc = upload(h, &h->d_phase, ph))) return rc;
  }
  (void)H;
  h->dirty = false;
  return FDOCT_OK;
}

// Bluestein tables for the +i transform of length n: X[k] = c[k] * sum_m (x[m] c[m]) conj(c[k-m]), c[m] = e^(+i pi m^2/n)
// (m^2 taken mod 2n in integers, so the angle stays exact); bhat = forward DFT of the wrapped conj(c), scaled by 1/Mb for
// the unscaled inverse transform that follows it in the kernels.  Computed in double.
void build_bluestein_tables(int n, int Mb, std::vector<float2>& chirp, std::vector<float2>& bhat) {
  std::vector<double> cr(n), ci(n);
  chirp.resize(n);
  for (long long m = 0; m < n; m++) {
    const double ang = kPi * (double)((m * m) % (2LL * n)) / (double)n;
    cr[m] = std::cos(ang);
    ci[m] = std::sin(ang);
    chirp[m] = make_float2((float)cr[m], (float)ci[m]);
  }
  std::vector<double> br(Mb, 0.0), bi(Mb, 0.0);
  for (int m = 0; m < n; m++) {
    br[m] = cr[m];
    bi[m] = -ci[m];
    if (m) {
      br[Mb - m] = cr[m];
      bi[Mb - m] = -ci[m];
    }
  }
  if (Mb & (Mb - 1)) {
    // Mb = 2^a 3^b 5^c (the LDS kernels' Bluestein, round 6: the smallest such length >= 2n - 1 instead of the next power of
    // two -- 2592 instead of 4096 around a 1283-point transform): the forward DFT by its definition, in double, with the
    // angle's index taken mod Mb in integers (<= 8192^2 complex multiply-adds, once per handle)
    std::vector<double> wr(Mb), wi(Mb);
    for (int j = 0; j < Mb; j++) {
      const double ang = -2.0 * kPi * (double)j / (double)Mb;
      wr[j] = std::cos(ang);
      wi[j] = std::sin(ang);
    }
    std::vector<int> nz;   // (the wrapped chirp has 2n - 1 non-zero entries)
    for (int m = 0; m < Mb; m++)
      if (br[m] != 0.0 || bi[m] != 0.0) nz.push_back(m);
    bhat.resize(Mb);
    for (int k = 0; k < Mb; k++) {
      double sr = 0.0, si = 0.0;
      for (int m : nz) {
        const int t = (int)(((long long)m * k) % Mb);
        sr += br[m] * wr[t] - bi[m] * wi[t];
        si += br[m] * wi[t] + bi[m] * wr[t];
      }
      bhat[k] = make_float2((float)(sr / Mb), (float)(si / Mb));
    }
    return;
  }
  // forward DFT of length Mb (power of two) in double: iterative radix-2
  for (int i = 1, j = 0; i < Mb; i++) {
    int bit = Mb >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      std::swap(br[i], br[j]);
      std::swap(bi[i], bi[j]);
    }
  }
  for (int len = 2; len <= Mb; len <<= 1) {
    const double ang = -2.0 * kPi / (double)len;
    for (int i = 0; i < Mb; i += len)
      for (int k = 0; k < len / 2; k++) {
        const double wr = std::cos(ang * k), wi = std::sin(ang * k);
        const double ur = br[i + k], ui = bi[i + k];
        const double vr = br[i + k + len / 2] * wr - bi[i + k + len / 2] * wi, vi = br[i + k + len / 2] * wi + bi[i + k + len / 2] * wr;
        br[i + k] = ur + vr;
        bi[i + k] = ui + vi;
        br[i + k + len / 2] = ur - vr;
        bi[i + k + len / 2] = ui - vi;
      }
  }
  bhat.resize(Mb);
  for (int m = 0; m < Mb; m++) bhat[m] = make_float2((float)(br[m] / Mb), (float)(bi[m] / Mb));
}

// Device tables of the generic path.
int rebuild_generic_state(fdoct_ctx* h) {
  int rc;
  const int W = h->W, N = h->N, MW = h->W * h->M;
  DEVICE_SCOPE(h);
  {
    std::vector<float> ib, il;
    reciprocal_words(scaled_copy(h->yb.v, plane_scales(h).yb), ib, il);
    if (h->yb.rows == 1) {
      if ((rc = upload(h, &h->d_ib, ib))) return rc;
      if ((rc = upload(h, &h->d_il, il))) return rc;
      if ((rc = dev_alloc(h, &h->d_ib2d, 0))) return rc;
      if ((rc = dev_alloc(h, &h->d_il2d, 0))) return rc;
    } else {
      if ((rc = upload(h, &h->d_ib2d, ib))) return rc;
      if ((rc = upload(h, &h->d_il2d, il))) return rc;
      if ((rc = dev_alloc(h, &h->d_ib, 0))) return rc;
      if ((rc = dev_alloc(h, &h->d_il, 0))) return rc;
    }
  }
  // (second words: what the float planes leave of the double ones -- read where the row is formed in double, BscanDark's band-pass)
  auto up_ref = [&](const RefFrame& f, double scale, float** d, float** d_lo) -> int {
    std::vector<float> t(f.v.size()), tl(f.v.size());
    for (size_t i = 0; i < t.size(); i++) {
      t[i] = (float)(f.v[i] * scale);
      tl[i] = (float)(f.v[i] * scale - (double)t[i]);
    }
    if (int e = upload(h, d_lo, tl)) return e;
    return upload(h, d, t);
  };
  if ((rc = up_ref(h->yp, plane_scales(h).yp, &h->d_yp, &h->d_yp_lo))) return rc;
  if ((rc = up_ref(h->yd, plane_scales(h).yd, &h->d_yd, &h->d_yd_lo))) return rc;
  std::vector<float> w(W), g(MW);
  for (int i = 0; i < W; i++) w[i] = (float)h->win[i];
  for (int i = 0; i < MW; i++) g[i] = (i < N) ? (float)h->frac[i] : 0.f;  // fractionalk[nearestkindex[q]], 0 past its end
  if ((rc = upload(h, &h->d_win_g, w))) return rc;
  {
    std::vector<float> wl(W);
    for (int i = 0; i < W; i++) wl[i] = (float)(h->win[i] - (double)w[i]);
    if ((rc = upload(h, &h->d_win_lo_g, wl))) return rc;
  }
  if ((rc = upload(h, &h->d_g_g, g))) return rc;
  if ((rc = upload(h, &h->d_idx_g, h->idx))) return rc;
  auto up_tw = [&](int n, float2** d) -> int {
    std::vector<float2> t(n);
    for (int j = 0; j < n; j++) {
      const double a = 2.0 * kPi * (double)j / (double)n;
      t[j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    return upload(h, d, t);
  };
  if ((rc = up_tw(N, &h->d_twg_n))) return rc;
  if ((N % 2) == 0 && (rc = up_tw(N / 2, &h->d_twg_nh))) return rc;
  if (h->blu_m) {
    const int n = generic_real_half(h) ? N / 2 : N, Mb = h->blu_m;
    std::vector<float2> chirp, bhat;
    build_bluestein_tables(n, Mb, chirp, bhat);
    if ((rc = upload(h, &h->d_blu_chirp, chirp))) return rc;
    if ((rc = upload(h, &h->d_blu_bhat, bhat))) return rc;
    if ((rc = up_tw(Mb, &h->d_twg_blu))) return rc;
  }
  if (h->M > 1 && h->zp_full) {  // the full-length zero-pad stage's two plans
    for (fdoct_ctx::GenericDftPlan* p : {&h->gzf, &h->gzi}) {
      if ((rc = up_tw(p->blu_m ? p->blu_m : p->n, &p->d_tw))) return rc;
      if (p->blu_m) {
        std::vector<float2> chirp, bhat;
        build_bluestein_tables(p->n, p->blu_m, chirp, bhat);
        if ((rc = upload(h, &p->d_chirp, chirp))) return rc;
        if ((rc = upload(h, &p->d_bhat, bhat))) return rc;
      }
    }
  }
  if (h->M > 1) {
    if ((rc = up_tw(W, &h->d_twg_w))) return rc;     // untangle factors of the half-length transforms
    if ((rc = up_tw(MW, &h->d_twg_mw))) return rc;
    if ((rc = up_tw(W / 2, &h->d_twg_wh))) return rc;
    if ((rc = up_tw(MW / 2, &h->d_twg_mwh))) return rc;
  }
  {
    std::vector<float2> ph(h->phase.size() / 2);
    for (size_t i = 0; i < ph.size(); i++) ph[i] = make_float2(h->phase[2 * i], h->phase[2 * i + 1]);
    if ((rc = upload(h, &h->d_phase, ph))) return rc;
  }
  if (!h->d_gen_tickets && (rc = dev_alloc(h, &h->d_gen_tickets, 64))) return rc;   // generic_kernel's row counters (launch_family_generic)
  h->dirty = false;
  h->generic_tables_ok = true;
  return FDOCT_OK;
}

// Tables of the wave-per-row kernels: packed gather sources and the twiddle blob
// [N/2 passes][M W/2 passes][W/2 passes][e^(2 pi i k/W), k < W/2][e^(2 pi i k/(M W)), k < W/2][e^(2 pi i k/N), k < D].
int rebuild_wave_state(fdoct_ctx* h) {
  // complex rows (dispersion phase): the final transform runs over the whole row, one gather source per point
  const bool cplx = !h->phase.empty();
  const int W = h->W, M = h->M, N = h->N, MW = W * M, NC = cplx ? N : N / 2, D = h->D;
  std::vector<uint32_t> gi(NC);
  auto src = [&](int q) -> uint32_t { return (q <= 0 || q >= N - 1) ? (uint32_t)MW : (uint32_t)h->idx[q]; };  // main:1164
  for (int n = 0; n < NC; n++) gi[n] = cplx ? src(n) : (src(2 * n) | (src(2 * n + 1) << 16));
  std::vector<float2> tw;
  auto unit = [&](double num, double den) {
    const double ang = 2.0 * kPi * num / den;
    return make_float2((float)std::cos(ang), (float)std::sin(ang));
  };
  auto pass_tables = [&](int n) {
    const WavePlan p = wave_plan(n);
    for (int i = 0; i < p.npass; i++)
      if (p.Ns[i] > 1)
        for (int k = 0; k < p.Ns[i]; k++) tw.push_back(unit((double)k, (double)p.Ns[i] * p.R[i]));
  };
  h->wave_off[0] = (int)tw.size();
  pass_tables(NC);
  h->wave_off[1] = (int)tw.size();
  if (M > 1) pass_tables(MW / 2);
  h->wave_off[2] = (int)tw.size();
  if (M > 1) pass_tables(W / 2);
  h->wave_off[3] = (int)tw.size();
  if (M > 1)
    for (int k = 0; k < W / 2; k++) tw.push_back(unit((double)k, (double)W));
  h->wave_off[4] = (int)tw.size();
  if (M > 1)
    for (int k = 0; k < W / 2; k++) tw.push_back(unit((double)k, (double)MW));
  h->wave_off[5] = (int)tw.size();
  // untangle factors of the real rows: bins below numdisplaypoints, or (displayed beyond N/2: the upper bins mirror) up to N/2
  if (!cplx)
    for (int k = 0; k < (D > N / 2 ? N / 2 + 1 : D); k++) tw.push_back(unit((double)k, (double)N));
  h->wave_tw_count = (int)tw.size();
  int rc;
  if ((rc = upload(h, &h->d_wave_gidx, gi))) return rc;
  if ((rc = upload(h, &h->d_wave_tw, tw))) return rc;
  h->wave_tables_ok = true;
  return FDOCT_OK;
}

}  // namespace fdoct_impl

// fdoct_route.cpp -- the dispatch of a call: choose_route decides everything before anything is launched, the
// passes in front of the chain run, one launcher per kernel family fills its argument block, finish_launch is the one epilogue;
// plus the long-row path's host side (plans, grouped launches) and the front end.  Part of the C-ABI layer (fdoct_ctx.h).
#include "fdoct_ctx.h"

namespace fdoct_impl {

int kernel_dtype(int dt) {
  switch (dt) {
    case FDOCT_U8: return FDOCT_K_U8;
    case FDOCT_U16: return FDOCT_K_U16;
    case FDOCT_F32: return FDOCT_K_F32;
    default: return -1;
  }
}

// medianBlur + INTER_AREA binning of device-resident raw frames into a packed, 16-byte-pitched buffer.
// Returns the binned frames in *out / *out_pitch (library workspace).
int run_frontend(fdoct_ctx* h, const void* d_raw, int kdt, int nframes, int raw_w, int raw_h, size_t raw_pitch, int mediann,
                 int binx, int biny, void** out, size_t* out_pitch) {
  if (kdt != FDOCT_K_U8 && kdt != FDOCT_K_U16) return fail(h, FDOCT_ERR_UNSUPPORTED, "the front end takes 8- or 16-bit camera frames");
  if (binx < 1 || biny < 1 || raw_w % binx || raw_h % biny) return fail(h, FDOCT_ERR_INVALID, "frame size must be a multiple of the bin factors");
  if (mediann != 0 && mediann != 3 && mediann != 5 && mediann != 7) return fail(h, FDOCT_ERR_INVALID, "mediann must be 0, 3, 5 or 7");
  // cv::medianBlur takes ksize 3 or 5 only for CV_16U (main:955 would throw): there is no reference behaviour to match
  if (mediann == 7 && kdt == FDOCT_K_U16) return fail(h, FDOCT_ERR_INVALID, "a 7x7 median exists for 8-bit frames only (cv::medianBlur)");
  const size_t es = kdt == FDOCT_K_U8 ? 1 : 2;
  int rc;
  hipStream_t st = h->stream;
  const void* src = d_raw;
  size_t src_pitch = raw_pitch;
  if (mediann > 0) {
    const size_t mp = ((size_t)raw_w * es + 15) & ~(size_t)15;
    if ((rc = dev_reserve(h, &h->ws_med, &h->ws_med_cap, mp * (size_t)raw_h * nframes))) return rc;
    HIP_TRY(h, launch_median(src, (long long)src_pitch, h->ws_med, (long long)mp, kdt, raw_w, raw_h, mediann, nframes, st));
    src = h->ws_med;
    src_pitch = mp;
  }
  const int ow = raw_w / binx, oh = raw_h / biny;
  const size_t op = ((size_t)ow * es + 15) & ~(size_t)15;
  if ((rc = dev_reserve(h, &h->ws_front, &h->ws_front_cap, op * (size_t)oh * nframes))) return rc;
  HIP_TRY(h, launch_bin(src, (long long)src_pitch, h->ws_front, (long long)op, kdt, ow, oh, binx, biny, nframes, st));
  *out = h->ws_front;
  *out_pitch = op;
  return FDOCT_OK;
}

// ---- long-row path (fdoct_big.hip) ----------------------------------------------------------------------------------
// The passes of an n-point transform (n = 2^a 3^b 5^c) as a few groups, each one launch with its data in LDS: the prime
// factors are dealt to G groups so that the groups' lengths come out as equal as they can (16384 = 128 x 128, 4096 = 64 x 64),
// G the smallest count that keeps every length within what a workgroup's tile holds.
bool big_plan_groups(int n, std::vector<fdoct_ctx::BigGroupPlan>& groups) {
  groups.clear();
  std::vector<int> primes;
  int m = n;
  for (int p : {5, 3, 2})
    while (m % p == 0) { primes.push_back(p); m /= p; }
  if (m != 1 || n < 2) return false;
  constexpr int kQmax = BIG_GROUP_TILE_VALUES / 8;   // 8 sub-problems of this many points fill the tile (64 contiguous bytes per element index)
  int G = 1;
  for (double cap = kQmax; cap < (double)n; cap *= kQmax) G++;
  for (; G <= 4; G++) {
    std::vector<long long> prod(G, 1);
    std::vector<std::vector<int>> fac(G);
    for (int p : primes) {  // largest factors first, each to the group that is shortest so far
      int best = 0;
      for (int g = 1; g < G; g++)
        if (prod[g] < prod[best]) best = g;
      prod[best] *= p;
      fac[best].push_back(p);
    }
    bool ok = true;
    for (int g = 0; g < G; g++) ok = ok && prod[g] <= BIG_GROUP_TILE_VALUES / 4;
    if (!ok) continue;
    long long P = 1;
    for (int g = 0; g < G; g++) {
      fdoct_ctx::BigGroupPlan gp;
      gp.P = (int)P;
      gp.Q = (int)prod[g];
      gp.F = (int)(n / (P * prod[g]));
      int twos = 0;
      for (int p : fac[g]) {
        if (p == 2) twos++;
        else gp.rad.push_back(p);
      }
      for (; twos >= 3; twos -= 3) gp.rad.push_back(8);
      if (twos == 2) gp.rad.push_back(4);
      if (twos == 1) gp.rad.push_back(2);
      if ((int)gp.rad.size() > BIG_GROUP_MAX_PASSES || gp.rad.empty()) { ok = false; break; }
      const long long S = (long long)gp.P * gp.F;
      int l2 = 4;
      while (l2 > 0 && (((long long)gp.Q << l2) > BIG_GROUP_TILE_VALUES || (1LL << l2) > S)) l2--;
      gp.log2ts = l2;
      groups.push_back(gp);
      P *= prod[g];
    }
    if (ok) return true;
    groups.clear();
  }
  return false;
}

// DFT plan of one length: Stockham radices when it factors into 2, 3, 5, else Bluestein around a power of two >= 2n - 1.
int big_plan_get(fdoct_ctx* h, int n, fdoct_ctx::BigPlan** out) {
  auto it = h->big_plans.find(n);
  if (it != h->big_plans.end()) {
    *out = &it->second;
    return FDOCT_OK;
  }
  fdoct_ctx::BigPlan p;
  auto radices = [](int len, std::vector<int>& rad) {  // 5s and 3s first, then 8s, then what is left of the power of two
    rad.clear();
    while (len % 5 == 0) { rad.push_back(5); len /= 5; }
    while (len % 3 == 0) { rad.push_back(3); len /= 3; }
    while (len % 8 == 0) { rad.push_back(8); len /= 8; }
    if (len % 4 == 0) { rad.push_back(4); len /= 4; }
    if (len % 2 == 0) { rad.push_back(2); len /= 2; }
    return len == 1;
  };
  int tn = n;
  if (!radices(n, p.rad)) {
    int mb = 1;
    while (mb < 2 * n - 1) mb <<= 1;
    p.mb = mb;
    radices(mb, p.rad);
    tn = mb;
    std::vector<float2> chirp, bhat;
    build_bluestein_tables(n, mb, chirp, bhat);
    int rc;
    if ((rc = upload(h, &p.d_chirp, chirp))) return rc;
    if ((rc = upload(h, &p.d_bhat, bhat))) return rc;
  }
  static const bool per_pass = [] { const char* e = std::getenv("FDOCT_BIG_PER_PASS"); return e && std::atoi(e) != 0; }();  // measurement: round 3's form
  if (!per_pass) big_plan_groups(tn, p.groups);
  std::vector<float2> tw(tn);
  for (int j = 0; j < tn; j++) {
    const double a = 2.0 * kPi * (double)j / (double)tn;
    tw[j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  int rc;
  if ((rc = upload(h, &p.d_tw, tw))) return rc;
  *out = &h->big_plans.emplace(n, p).first->second;
  return FDOCT_OK;
}

void big_plans_free(fdoct_ctx* h) {
  for (auto& kv : h->big_plans)
    for (float2* p : {kv.second.d_tw, kv.second.d_chirp, kv.second.d_bhat})
      if (p) (void)hipFree(p);
  h->big_plans.clear();
}

// What stands in front of a transform, fused into the loads of its first launch (or run as a kernel of its own where the
// transform is not one of grouped launches): the row of floats read as complex (A4's first transform), the W-point spectrum
// re-packed into the M W-point one (A4), the slope step and lambda -> k gather with the phase (A5 / A6 / A6').
struct BigLoader {
  int load = BIG_LOAD_CPLX;
  const float* yr = nullptr;
  const float2* yc = nullptr;
  const float2* spec = nullptr;
  int ylen = 0, W = 0, bandpass = 0;
  const int32_t* idx = nullptr;
  const float* g = nullptr;
  const float2* phase = nullptr;
};

// X = IDFT_n (+i exponent, unscaled) of `rows` rows; x holds them (ld == null) or is free and the loader supplies them;
// `other` is the second buffer (both hold rows * max(n, mb) values).  *result = the buffer that holds the rows * n result;
// out_limit > 0: only the first out_limit values of each result row are needed (and, with grouped launches, written).
int big_idft(fdoct_ctx* h, float2* x, float2* other, long long rows, int n, float2** result, hipStream_t st, const BigLoader* ld = nullptr,
             int out_limit = 0) {
  fdoct_ctx::BigPlan* p = nullptr;
  int rc;
  if ((rc = big_plan_get(h, n, &p))) return rc;
  auto materialise = [&]() -> int {  // the loader as a kernel of its own, into x
    if (!ld) return FDOCT_OK;
    switch (ld->load) {
      case BIG_LOAD_REAL: HIP_TRY(h, big_launch_real_to_complex(ld->yr, rows * n, x, st)); break;
      case BIG_LOAD_PAD: HIP_TRY(h, big_launch_pad(ld->spec, rows, ld->W, n, ld->bandpass, x, st)); break;
      case BIG_LOAD_RESAMPLE: HIP_TRY(h, big_launch_resample(ld->yr, ld->yc, rows, ld->ylen, n, ld->idx, ld->g, ld->phase, x, st)); break;
      default: break;
    }
    return FDOCT_OK;
  };
  // the passes of one len-point transform over src -> ... -> *last (ping-pong between the two buffers)
  auto passes = [&](float2*& src, float2*& dst, int len, const BigLoader* first_ld, int limit) -> int {
    if (!p->groups.empty()) {
      for (size_t gi = 0; gi < p->groups.size(); gi++) {
        const auto& gp = p->groups[gi];
        BigGroup g{};
        g.src = src; g.dst = dst; g.rows = rows; g.n = len;
        g.P = gp.P; g.Q = gp.Q; g.F = gp.F; g.log2ts = gp.log2ts;
        g.npass = (int)gp.rad.size();
        for (int i = 0; i < g.npass; i++) g.rad[i] = gp.rad[i];
        g.out_limit = (gi + 1 == p->groups.size() && limit > 0) ? limit : len;
        g.tw = p->d_tw;
        g.load = BIG_LOAD_CPLX;
        if (gi == 0 && first_ld) {
          g.load = first_ld->load;
          g.yr = first_ld->yr; g.yc = first_ld->yc; g.ylen = first_ld->ylen;
          g.idx = first_ld->idx; g.g = first_ld->g; g.phase = first_ld->phase;
          g.W = first_ld->W; g.bandpass = first_ld->bandpass;
          if (first_ld->load == BIG_LOAD_PAD) g.src = first_ld->spec;
        }
        HIP_TRY(h, big_launch_fft_group(g, st));
        std::swap(src, dst);
      }
      return FDOCT_OK;
    }
    int Ns = 1;
    for (int R : p->rad) {
      HIP_TRY(h, big_launch_fft_pass(src, dst, rows, len, R, Ns, p->d_tw, st));
      std::swap(src, dst);
      Ns *= R;
    }
    return FDOCT_OK;
  };
  float2 *src = x, *dst = other;
  if (!p->mb) {
    const bool fused = ld && !p->groups.empty();
    if (!fused && (rc = materialise())) return rc;
    if ((rc = passes(src, dst, n, fused ? ld : nullptr, out_limit))) return rc;
    *result = src;
    return FDOCT_OK;
  }
  // Bluestein: u = conj(x c) zero-padded; conj(IDFT u) = DFT(x c); times bhat; IDFT; times c
  if ((rc = materialise())) return rc;
  HIP_TRY(h, big_launch_chirp_in(x, rows, n, p->mb, p->d_chirp, other, st));
  src = other;
  dst = x;
  if ((rc = passes(src, dst, p->mb, nullptr, 0))) return rc;
  HIP_TRY(h, big_launch_conj_mul(src, rows, p->mb, p->d_bhat, st));
  if ((rc = passes(src, dst, p->mb, nullptr, 0))) return rc;
  HIP_TRY(h, big_launch_chirp_out(src, rows, n, p->mb, p->d_chirp, dst, st));
  *result = dst;
  return FDOCT_OK;
}

// The whole chain for device-resident frames on the long-row path, chunk by chunk of whole averaging groups.
int run_big(fdoct_ctx* h, const void* kframes, const float* kframes_lo, int kdt, size_t kpitch, int nframes, bool need_minmax, float* k_mag, float* k_db,
            hipStream_t st) {
  const int W = h->W, H = h->H, N = h->N, D = h->D, M = h->M, A = h->A;
  // the padded spectrum / upsampled row: W + 2 floor((M W - W) / 2) points (main:229) -- M W, or M W - 1 for an odd width under an
  // even multiplier
  const int MW = W + 2 * ((W * M - W) / 2);
  int rc;
  size_t lmax = (size_t)std::max(N, M > 1 ? MW : 0);
  for (int n : {N, M > 1 ? W : 0, M > 1 ? MW : 0}) {
    if (!n) continue;
    fdoct_ctx::BigPlan* p = nullptr;
    if ((rc = big_plan_get(h, n, &p))) return rc;
    lmax = std::max(lmax, (size_t)std::max(n, p->mb));
  }
  const size_t per_group = (size_t)A * H * ((size_t)W * 4 + 2 * lmax * sizeof(float2));
  long long cg = (long long)(((size_t)2 << 30) / per_group);
  const int G = nframes / A;
  if (cg < 1) cg = 1;
  if (cg > G) cg = G;
  const size_t crow = (size_t)cg * A * H;
  if ((rc = dev_reserve(h, &h->ws_big_y, &h->ws_big_y_cap, crow * W * 4))) return rc;
  if ((rc = dev_reserve(h, &h->ws_big_a, &h->ws_big_a_cap, crow * lmax * sizeof(float2)))) return rc;
  if ((rc = dev_reserve(h, &h->ws_big_b, &h->ws_big_b_cap, crow * lmax * sizeof(float2)))) return rc;
  for (long long g0 = 0; g0 < G; g0 += cg) {
    const long long ng = std::min<long long>(cg, G - g0);
    BigArgs a{};
    a.frames = static_cast<const unsigned char*>(kframes) + (size_t)g0 * A * H * kpitch;
    a.frames_lo = kframes_lo ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(kframes_lo) + (size_t)g0 * A * H * kpitch) : nullptr;
    a.pitch_bytes = (long long)kpitch;
    a.in_rows = ng * A * H;
    a.out_rows = ng * H;
    a.dtype = kdt;
    a.W = W; a.H = H; a.N = N; a.D = D; a.M = M; a.A = A;
    a.ib = h->yb.rows == 1 ? h->d_ib : h->d_ib2d;
    a.il = h->yb.rows == 1 ? h->d_il : h->d_il2d;
    a.ib_2d = h->yb.rows > 1;
    a.yp = h->d_yp; a.yp_2d = h->yp.rows > 1;
    a.yd = h->d_yd; a.yd_2d = h->yd.rows > 1;
    a.win = h->d_win_g;
    a.minmax = need_minmax ? h->d_minmax + (size_t)g0 * A : nullptr;
    a.rowwisenormalize = h->cfg.rowwisenormalize;
    a.dcmask = h->cfg.dc_mask;
    a.inv_A = (float)(1.0 / (double)A);
    a.eps = (h->cfg.variant == FDOCT_VARIANT_SIM) ? 1e-6f : 1e-5f;
    a.db_scale = (float)(20.0 / 2.303 * 0.6931471805599453);
    HIP_TRY(h, big_launch_pre(a, h->ws_big_y, st));
    // Buffer discipline of big_idft with a loader: the first launch reads the loader's source and writes `other`, the next one
    // writes `x`, and so on; so the source may live in x (it is dead once the first launch is through) but never in `other`.
    // A transform that cannot fuse its loader (one launch per pass, Bluestein) materialises the rows into x first: there the
    // source must not live in x.
    float2 *bufa = h->ws_big_a, *bufb = h->ws_big_b, *res = nullptr;
    auto fuses = [&](int n, bool* yes) -> int {
      fdoct_ctx::BigPlan* p = nullptr;
      if (int rc2 = big_plan_get(h, n, &p)) return rc2;
      *yes = !p->mb && !p->groups.empty();
      return FDOCT_OK;
    };
    BigLoader rs;                 // A5 / A6 / A6': what the final transform reads
    rs.load = BIG_LOAD_RESAMPLE;
    rs.yr = h->ws_big_y;
    rs.ylen = W;
    rs.idx = h->d_idx_g;
    rs.g = h->d_g_g;
    rs.phase = h->d_phase;
    float2* held = nullptr;       // the buffer the final transform's source rows live in (null: the float rows)
    if (M > 1) {  // A4
      BigLoader l1;
      l1.load = BIG_LOAD_REAL;
      l1.yr = h->ws_big_y;
      if ((rc = big_idft(h, bufa, bufb, a.in_rows, W, &res, st, &l1))) return rc;
      BigLoader l2;
      l2.load = BIG_LOAD_PAD;
      l2.spec = res;
      l2.W = W;
      l2.bandpass = h->bandpass ? 1 : 0;
      float2* spare = (res == bufa) ? bufb : bufa;
      bool f = false;
      if ((rc = fuses(MW, &f))) return rc;
      if ((rc = f ? big_idft(h, res, spare, a.in_rows, MW, &res, st, &l2) : big_idft(h, spare, res, a.in_rows, MW, &res, st, &l2))) return rc;
      rs.yr = nullptr;
      rs.yc = res;
      rs.ylen = MW;
      held = res;
    }
    {  // A5 / A6 / A7; only the first numdisplaypoints bins of the result are needed
      float2* spare = held ? (held == bufa ? bufb : bufa) : bufb;
      float2* mine = held ? held : bufa;
      bool f = false;
      if ((rc = fuses(N, &f))) return rc;
      if ((rc = f ? big_idft(h, mine, spare, a.in_rows, N, &res, st, &rs, D) : big_idft(h, held ? spare : mine, held ? mine : spare, a.in_rows, N, &res, st, &rs, D)))
        return rc;
    }
    HIP_TRY(h, big_launch_post(res, a, k_mag ? k_mag + (size_t)g0 * H * D : nullptr, k_db ? k_db + (size_t)g0 * H * D : nullptr, st));
  }
  return FDOCT_OK;
}

// Can the chain write the reference's D x H layout itself (fused_kernel's TRO instantiations)?  The acquisition
// configurations on the 1024-point row-swap plan: 8/16-bit frames that go to the kernel as they are, 1-row or full-frame
// background, none or the whole-frame normalisation, rows in fours and depth bins in whole write-out steps, 16-byte
// aligned outputs.
bool fused_transposed_store_applies(const fdoct_ctx* h, fdoct_dtype dtype, const void* d_frames, size_t pitch_bytes,
                                    const float* d_out_bscan, const float* d_out_db, int nframes) {
  if (!h->tro_enabled || h->use_generic || h->staged || h->force_general || h->cplx) return false;
  const FusedPlan& p = h->plan;
  if (!fused_tro_compiled(p.kind, p.T, p.WCH)) return false;
  if (dtype != FDOCT_U8 && dtype != FDOCT_U16) return false;
  if (h->fe_median > 0 || h->fe_binx > 1 || h->fe_biny > 1 || h->cfg.movavgn > 0) return false;
  if (h->W != 8 * p.T * p.WCH || !h->yb.rows || h->yp.rows || h->yd.rows || h->cfg.rowwisenormalize) return false;
  const bool normalize = (h->cfg.variant == FDOCT_VARIANT_SIM) || !h->cfg.donotnormalize;
  if ((normalize || h->yb.rows > 1) && h->A != 1) return false;  // (those instantiations exist for one frame per B-scan)
  if (p.kind != 1 && (normalize || h->yb.rows > 1)) return false;  // (... and for the row-swap plan only: the 512-point plan has the plain and the averaging kernel)
  const size_t es = dtype == FDOCT_U8 ? 1 : 2, valign = dtype == FDOCT_U8 ? 8 : 16;
  const size_t pitch = pitch_bytes ? pitch_bytes : es * (size_t)h->W;
  if (((uintptr_t)d_frames % valign) || (pitch % valign)) return false;
  if ((h->H % 4) || (h->D % fused_tro_step_bins(64 / p.T)) || h->D > h->NC) return false;
  // (both words: a full-frame background brings its second word along with the prefetched row -- no LDS plane; a 1-row one needs
  // the plane next to the ring, which then holds one computing wave less)
  if (h->precise_div && h->yb.rows > 1 && !fused_il_half(true, p.WCH)) return false;
  const int rpw = 64 / p.T;
  if (rpw == 4) {
    // four rows per wave: no ring -- groups of four waves own tiles and write them out from their own row buffers (fused_kernel,
    // TRO_INPLACE): at least one group must fit, and a launch override must leave whole groups
    if ((size_t)160 * 1024 - 64 < tro_const_lds_bytes(h, (int)es, normalize) + (size_t)fused_tro_group_waves() * h->scratch_bytes * rpw) return false;
    if (fused_max_block(h->NC, p.T, true, p.kind) / 64 < fused_tro_group_waves()) return false;
    if (h->block_override && h->block_override / 64 < fused_tro_group_waves()) return false;
  } else if (fused_tro_ring_pick((size_t)160 * 1024 - 64 - tro_const_lds_bytes(h, (int)es, normalize) - (size_t)h->scratch_bytes * rpw, h->D, rpw) == 0) {
    return false;  // (no ring next to one computing wave)
  }
  if (((uintptr_t)d_out_bscan % 16) || ((uintptr_t)d_out_db % 16)) return false;
  if ((long long)(nframes / h->A) * h->H >= 0x7fffffffLL) return false;
  // the write-out addresses one B-scan with 32-bit byte offsets inside a buffer descriptor of 0x7ffffff0 bytes
  if ((size_t)h->D * (size_t)h->H * 4 >= 0x7ffffff0u) return false;
  return true;
}

// Decides the route of a call.  frames_addr / pitch_bytes / out addresses: as the caller gave them (fdoct_prepare: an aligned,
// packed set-up).  May rebuild device tables and compile (hipRTC) -- never launches.
int choose_route(fdoct_ctx* h, fdoct_dtype dtype, uintptr_t frames_addr, size_t pitch_bytes, uintptr_t out_bscan_addr,
                 uintptr_t out_db_addr, fdoct_layout layout, int nframes, Route* r) {
  int rc;
  if (h->dirty && (rc = rebuild_device_state(h))) return rc;
  if (h->D > h->N) return fail(h, FDOCT_ERR_INVALID, "numdisplaypoints > numfftpoints");
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  const long long out_rows = (long long)(nframes / A) * H;
  const size_t es = dtype_size(dtype);
  *r = Route{};
  h->jit_note.clear();   // (the note describes THIS call's route: a refused run-time compile, or the long-row path)
  int kdt = kernel_dtype(dtype);
  const bool normalize = (h->cfg.variant == FDOCT_VARIANT_SIM) || !h->cfg.donotnormalize;
  // with row-wise normalisation on, every non-degenerate row already spans [0,1] and the whole-frame pass (main:1128) is the identity
  r->need_minmax = normalize && !h->cfg.rowwisenormalize;
  // pi / dark frames, the band-pass and the normalisations are compile-time options of the wave-per-row kernel: the library's own
  // instantiations are the plain set-up, a handle that uses one of them gets its kernel from the run-time compiler
  // ... and so are the dispersion phase (complex rows, a full-length final transform) and a display beyond numfftpoints / 2
  const int wave_opt = (h->yp.rows ? FDOCT_WAVE_OPT_PI : 0) | (h->yd.rows ? FDOCT_WAVE_OPT_DARK : 0) |
                       (h->bandpass && h->M > 1 ? FDOCT_WAVE_OPT_BANDPASS : 0) | (h->cfg.rowwisenormalize ? FDOCT_WAVE_OPT_ROWNORM : 0) |
                       (r->need_minmax ? FDOCT_WAVE_OPT_FRAMENORM : 0) | (!h->phase.empty() ? FDOCT_WAVE_OPT_CPLX : 0) |
                       (h->phase.empty() && D > h->N / 2 ? FDOCT_WAVE_OPT_DEEP : 0);
  auto wave_tables = [&]() -> int {  // (the wave tables read the resample table's device copies)
    if (!h->generic_tables_ok) {
      const bool keep = h->use_generic;
      int rc2 = select_generic(h);
      h->use_generic = keep;
      if (rc2) return rc2;
      if ((rc2 = rebuild_generic_state(h))) return rc2;
    }
    return h->wave_tables_ok ? FDOCT_OK : rebuild_wave_state(h);
  };
  // 2 x 2 binning with nothing else in front of the chain, on a configuration the wave-per-row kernel takes: the kernel compiled
  // for the handle does the binning in its own loads (FDOCT_WAVE_OPT_BIN2) and the pass over the raw frames is skipped.
  // (Measured on the shipped shapes, tools/bench_generic.py with and without FDOCT_JIT=0: + 4.5 % on 160-sample 8-bit rows, + 5 % on
  // 640-sample 16-bit rows, - 1 % on 640-sample 8-bit rows -- twenty 2-byte loads per lane cost what the pass saves: those keep the pass.)
  if (h->fe_median == 0 && h->fe_binx == 2 && h->fe_biny == 2 && (dtype == FDOCT_U16 || (dtype == FDOCT_U8 && W <= 320)) && h->cfg.movavgn == 0 &&
      h->jit && h->use_generic && !h->use_big && h->plan_override > -2 && h->phase.empty() && D <= h->N / 2 && !r->need_minmax &&
      (frames_addr % 4 == 0) && (pitch_bytes % 4 == 0) && pitch_bytes >= es * 2 * (size_t)W && out_rows < 0x7fffffffLL &&
      wave_jit_shape_ok(W, h->M, h->N, D)) {
    if ((rc = wave_tables())) return rc;
    const size_t shared = wave_shared_lds_bytes(h->wave_tw_count, W, h->M, h->N, h->yb.rows > 1);
    std::string why;
    hipFunction_t fn = nullptr;
    if (shared + wave_private_lds_bytes(W, h->M, h->N) > 160 * 1024 - 64) {
      // no room for even one wave: the ordinary path (binning pass, then whichever kernel fits)
    } else if (wave_jit_get(W, h->M, h->N, kdt, D, wave_opt | FDOCT_WAVE_OPT_BIN2, h->device, &fn, &why) == hipSuccess) {
      r->bin2_in_kernel = true;
      r->jit_fn = fn;
      r->wave_opt = wave_opt | FDOCT_WAVE_OPT_BIN2;
    }
    h->jit_note = why;
  }
  // ---- passes in front of the chain, and what they leave for its kernel to read
  uintptr_t kaddr = frames_addr;
  size_t kpitch = pitch_bytes;
  if (!r->bin2_in_kernel && (h->fe_median > 0 || h->fe_binx > 1 || h->fe_biny > 1)) {
    if (dtype != FDOCT_U8 && dtype != FDOCT_U16)
      return fail(h, FDOCT_ERR_UNSUPPORTED, "the front end (median / binning) takes the camera's 8- or 16-bit frames");
    if (pitch_bytes < es * (size_t)W * h->fe_binx) return fail(h, FDOCT_ERR_INVALID, "pitch smaller than a raw camera row");
    r->frontend = true;
    kaddr = 0;                                                  // a library workspace: aligned
    kpitch = ((size_t)W * es + 15) & ~(size_t)15;
  }
  if (dtype == FDOCT_F64) {
    if (pitch_bytes % 8) return fail(h, FDOCT_ERR_INVALID, "f64 pitch must be a multiple of 8");
    r->narrow_f64 = true;
    kaddr = 0;
    kpitch = (size_t)W * 4;
    kdt = FDOCT_K_F32;
  }
  if (h->cfg.movavgn > 0) {
    r->movavg = true;
    // (float frames: the tap sums of non-integer samples go on as two planes, like the doubles' -- round 6: a float sum rounds at
    // the size of the DC level, 0.87 x the tolerance from the chain in double under fringes of 1e-3 of it in the sweeps)
    r->mov_lo = dtype == FDOCT_F32 && !r->frontend && (pitch_bytes % 4 == 0) && (frames_addr % 4 == 0);
    kaddr = 0;
    kpitch = (size_t)W * 4;
    kdt = FDOCT_K_F32;
  }
  r->kdt = kdt;
  r->kpitch = kpitch;
  // the specialised kernels read 16-byte vectors; anything else goes through the generic kernel
  const size_t valign = (kdt == FDOCT_K_U8) ? 8 : 16;
  const bool misaligned = (kaddr % valign) || (kpitch % valign);
  const bool run_generic = h->use_generic || misaligned;
  if (run_generic && !h->generic_tables_ok) {
    const bool keep = h->use_generic;
    rc = select_generic(h);
    h->use_generic = keep;
    if (rc) return rc;
    if ((rc = rebuild_generic_state(h))) return rc;
  }
  if (run_generic && h->staged) return fail(h, FDOCT_ERR_UNSUPPORTED, "staged mode needs a specialised kernel for this configuration");

  const bool transposed = layout == FDOCT_LAYOUT_TRANSPOSED_DxH;
  r->tro = transposed && !run_generic && !r->frontend && !r->narrow_f64 && !r->movavg &&
           fused_transposed_store_applies(h, dtype, reinterpret_cast<const void*>(frames_addr), pitch_bytes,
                                          reinterpret_cast<const float*>(out_bscan_addr), reinterpret_cast<const float*>(out_db_addr), nframes);
  r->transpose_pass = transposed && !r->tro;

  // the acquisition configurations the reference ships: one wave per A-scan (fdoct_wave.hip) instead of one workgroup
  // (frames handed over as doubles carry a low word per sample: the fused any-option, workgroup-per-row and long-row kernels take it)
  const bool wave_scope = run_generic && !h->use_big && h->plan_override > -2 && kdt >= 0 && !r->narrow_f64 && !r->mov_lo &&
                          (kaddr % 4 == 0) && (kpitch % 4 == 0) && out_rows < 0x7fffffffLL;
  if (r->bin2_in_kernel && !wave_scope) return fail(h, FDOCT_ERR_DEVICE, "internal: binning left to a kernel that does not run");
  bool run_wave = r->bin2_in_kernel;
  bool wave_builtin = false;
  if (wave_scope && !run_wave) {
    wave_builtin = wave_opt == 0 && wave_kernel_available(W, h->M, h->N, kdt, D);
    // any other shape the template can take: compiled for this handle's geometry at run time (fdoct_set_jit); the first call
    // (or fdoct_prepare) pays the compile, a refusal falls back to the workgroup-per-row kernel
    if (!wave_builtin && h->jit && wave_jit_shape_ok(W, h->M, h->N, D, wave_opt)) {
      std::string why;
      hipFunction_t fn = nullptr;
      if (wave_jit_get(W, h->M, h->N, kdt, D, wave_opt, h->device, &fn, &why) == hipSuccess) {
        r->jit_fn = fn;
        r->wave_opt = wave_opt;
      }
      h->jit_note = why;
    }
    run_wave = wave_builtin || r->jit_fn;
  }
  if (run_wave) {
    if ((rc = wave_tables())) return rc;
    const size_t shared = wave_shared_lds_bytes(h->wave_tw_count, W, h->M, h->N, h->yb.rows > 1, r->wave_opt);
    if (shared + wave_private_lds_bytes(W, h->M, h->N, r->wave_opt) > 160 * 1024 - 64) {  // not even one wave's buffer next to the tables
      if (r->bin2_in_kernel) return fail(h, FDOCT_ERR_DEVICE, "internal: binning left to a kernel that did not launch");
      run_wave = false;
      r->jit_fn = nullptr;
    }
  }
  if (run_wave)
    r->family = r->jit_fn ? FDOCT_KERNEL_WAVE_JIT : FDOCT_KERNEL_WAVE;
  else if (run_generic) {
    r->family = h->use_big ? FDOCT_KERNEL_LONG_ROWS : FDOCT_KERNEL_GENERIC;
    // the cliff an integrator should see: rows in HBM run an order of magnitude below their LDS-resident neighbours
    if (h->use_big)
      h->jit_note = "this geometry runs on the long-row path (rows in HBM, fdoct_big.hip: one launch per group of DFT passes): its transforms do not fit "
                    "the 160 KB of LDS of a compute unit; expect 1e6 ... 1e7 A-scans/s where LDS-resident rows reach 1e8";
  }
  else
    r->family = h->staged ? FDOCT_KERNEL_FUSED_STAGED : (r->tro ? FDOCT_KERNEL_FUSED_TRANSPOSED : FDOCT_KERNEL_FUSED);
  return FDOCT_OK;
}

// After the chain's kernel(s) of any family: end-of-kernel event, the transpose pass where the chain did not write D x H itself,
// end-of-call event, and the call's figures for fdoct_get_timing.
int finish_launch(fdoct_ctx* h, const Route& r, const Call& c, bool staged_timing) {
  hipStream_t st = c.st;
  if (h->record_now && h->rec_last) HIP_TRY(h, hipEventRecord(h->ev[2], st));
  if (r.transpose_pass) {
    if (c.d_out_bscan) HIP_TRY(h, launch_transpose(c.k_mag, c.d_out_bscan, h->H, h->D, c.G, st));
    if (c.d_out_db) HIP_TRY(h, launch_transpose(c.k_db, c.d_out_db, h->H, h->D, c.G, st));
  }
  if (h->record_now && h->rec_last) HIP_TRY(h, hipEventRecord(h->ev[3], st));
  h->last_kernel = r.family;
  h->timing.ascans = (uint64_t)c.in_rows;
  h->timing.bytes_in = (uint64_t)c.in_rows * h->W * c.es;
  h->timing.bytes_out = (uint64_t)c.out_rows * h->D * 4 * ((c.d_out_bscan ? 1 : 0) + (c.d_out_db ? 1 : 0));
  h->timing_pending = h->record_now;
  h->timing_staged = staged_timing;
  return FDOCT_OK;
}

float chain_eps(const fdoct_ctx* h) { return (h->cfg.variant == FDOCT_VARIANT_SIM) ? 1e-6f : 1e-5f; }  // sim:949 / main:1222
constexpr float kDbScale = (float)(20.0 / 2.303 * 0.6931471805599453);                                // main:1236, times ln 2 (the kernels use log2)

int launch_family_wave(fdoct_ctx* h, const Route& r, const Call& c) {
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  WaveArgs wa{};
  wa.frames = c.kframes;
  wa.pitch_bytes = (long long)r.kpitch;
  wa.total_out_rows = c.out_rows;
  wa.dtype = r.kdt;
  wa.H = H; wa.D = D; wa.A = A;
  wa.ib = h->yb.rows == 1 ? h->d_ib : h->d_ib2d;
  wa.il = h->yb.rows == 1 ? h->d_il : h->d_il2d;
  wa.ib_2d = h->yb.rows > 1;
  wa.win = h->d_win_g;
  wa.win_lo = h->d_win_lo_g;
  wa.yp_lo = h->d_yp_lo; wa.yd_lo = h->d_yd_lo;
  wa.g = h->d_g_g;
  wa.gidx = h->d_wave_gidx;
  wa.tw = h->d_wave_tw;
  wa.tw_count = h->wave_tw_count;
  wa.off_nc = h->wave_off[0]; wa.off_lh = h->wave_off[1]; wa.off_wh = h->wave_off[2];
  wa.off_tww = h->wave_off[3]; wa.off_twmw = h->wave_off[4]; wa.off_twn = h->wave_off[5];
  wa.dcmask = h->cfg.dc_mask;
  wa.inv_A = (float)(1.0 / (double)A);
  wa.eps = chain_eps(h);
  wa.db_scale = kDbScale;
  wa.out_mag = c.k_mag;
  wa.out_db = c.k_db;
  wa.yp = h->d_yp; wa.yp_2d = h->yp.rows > 1;
  wa.yd = h->d_yd; wa.yd_2d = h->yd.rows > 1;
  wa.minmax = r.need_minmax ? h->d_minmax : nullptr;
  wa.phase = h->d_phase;
#ifdef FDOCT_WAVE_PROBE  // measurement build: per-phase cycles of the first workgroups' waves, printed every 50 calls
  {
    static unsigned long long* d_probe = nullptr;
    const size_t pbytes = 4 * 16 * 12 * 8;
    if (!d_probe) {
      (void)hipMalloc(reinterpret_cast<void**>(&d_probe), pbytes);
      (void)hipMemset(d_probe, 0, pbytes);
    }
    wa.probe = d_probe;
    static int calls = 0;
    if (++calls % 50 == 0) {
      std::vector<unsigned long long> v(4 * 16 * 12);
      (void)hipStreamSynchronize(c.st);
      (void)hipMemcpy(v.data(), d_probe, pbytes, hipMemcpyDeviceToHost);
      static const char* names[9] = {"load+A2/A3", "fwd W/2", "re-pack", "inv MW/2", "slope", "gather", "final N/2", "untangle", "epilogue"};
      double tot = 0;
      double sum[9] = {};
      int nw = 0;
      for (int w = 0; w < 64; w++) {
        if (!v[w * 12 + 6]) continue;
        nw++;
        for (int i = 0; i < 9; i++) sum[i] += (double)v[w * 12 + i];
      }
      for (int i = 0; i < 9; i++) tot += sum[i];
      std::fprintf(stderr, "[wave probe] %d waves:", nw);
      for (int i = 0; i < 9; i++) std::fprintf(stderr, " %s %.1f%%", names[i], 100.0 * sum[i] / (tot > 0 ? tot : 1));
      std::fprintf(stderr, "\n");
    }
  }
#endif
  const size_t shared = wave_shared_lds_bytes(wa.tw_count, W, h->M, h->N, wa.ib_2d != 0, r.wave_opt);
  const size_t priv = wave_private_lds_bytes(W, h->M, h->N, r.wave_opt);
  int waves = (int)((160 * 1024 - 64 - shared) / priv);  // >= 1: choose_route
  if (waves > wave_max_waves(W, h->M, h->N, r.wave_opt)) waves = wave_max_waves(W, h->M, h->N, r.wave_opt);
  if (h->block_override && h->block_override / 64 >= 1 && h->block_override / 64 <= waves) waves = h->block_override / 64;
  long long wgrid = h->num_cu;
  const int rpw_ = wave_rows_per_wave(W, h->M, h->N, r.wave_opt);   // (two rows per wave on the short zero-padded shapes)
  const long long need = (c.out_rows + (long long)waves * rpw_ - 1) / ((long long)waves * rpw_);
  if (h->grid_override > 0) wgrid = h->grid_override;
  if (wgrid > need) wgrid = need;
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[1], c.st));
  if (r.jit_fn)
    HIP_TRY(h, wave_jit_launch(r.jit_fn, wa, (int)wgrid, waves, shared + (size_t)waves * priv, c.st));
  else
    HIP_TRY(h, launch_wave(W, h->M, h->N, wa, (int)wgrid, waves, shared + (size_t)waves * priv, c.st));
  return finish_launch(h, r, c, false);
}

int launch_family_long_rows(fdoct_ctx* h, const Route& r, const Call& c) {
  int rc;
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[1], c.st));
  if ((rc = run_big(h, c.kframes, c.kframes_lo, r.kdt, r.kpitch, c.nframes, r.need_minmax, c.k_mag, c.k_db, c.st))) return rc;
  return finish_launch(h, r, c, false);
}

int launch_family_generic(fdoct_ctx* h, const Route& r, const Call& c) {
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  GenericArgs ga{};
  ga.frames = c.kframes;
  ga.frames_lo = c.kframes_lo;
  ga.pitch_bytes = (long long)r.kpitch;
  ga.total_out_rows = c.out_rows;
  ga.dtype = r.kdt;
  ga.W = W; ga.H = H; ga.N = h->N; ga.D = D; ga.M = h->M; ga.A = A;
  ga.L = generic_buffer_len(h);
  ga.real_half = generic_real_half(h) ? 1 : 0;
  ga.ybuf_len = (W + 3) & ~3;
  ga.ib = h->yb.rows == 1 ? h->d_ib : h->d_ib2d;
  ga.il = h->yb.rows == 1 ? h->d_il : h->d_il2d;
  ga.ib_2d = h->yb.rows > 1;
  ga.yp = h->d_yp; ga.yp_2d = h->yp.rows > 1;
  ga.yd = h->d_yd; ga.yd_2d = h->yd.rows > 1;
  ga.win = h->d_win_g;
  ga.win_lo = h->d_win_lo_g;
  ga.yp_lo = h->d_yp_lo; ga.yd_lo = h->d_yd_lo;
  ga.g = h->d_g_g;
  ga.idx = h->d_idx_g;
  ga.phase = h->d_phase;
  ga.minmax = r.need_minmax ? h->d_minmax : nullptr;
  ga.tw_n = h->d_twg_n; ga.tw_nh = h->d_twg_nh; ga.tw_w = h->d_twg_w; ga.tw_mw = h->d_twg_mw;
  ga.tw_wh = h->d_twg_wh; ga.tw_mwh = h->d_twg_mwh;
  auto put_plan = [](const std::vector<int>& rad, int* rr, unsigned* mag) {
    unsigned long long ns = 1;
    for (size_t i = 0; i < rad.size(); i++) {
      rr[i] = rad[i];
      mag[i] = (unsigned)(((1ull << 32) + ns - 1) / ns);  // ceil(2^32 / Ns); unused for Ns == 1
      ns *= (unsigned)rad[i];
    }
  };
  put_plan(h->rad_n, ga.rad_n, ga.mag_n);
  if (ga.real_half) put_plan(h->rad_nh, ga.rad_nh, ga.mag_nh);
  ga.npass_nh = (int)h->rad_nh.size();
  put_plan(h->rad_wh, ga.rad_wh, ga.mag_wh);
  put_plan(h->rad_mwh, ga.rad_mwh, ga.mag_mwh);
  ga.npass_n = (int)h->rad_n.size(); ga.npass_wh = (int)h->rad_wh.size(); ga.npass_mwh = (int)h->rad_mwh.size();
  ga.blu_m = h->blu_m;
  if (h->blu_m) {
    put_plan(h->rad_blu, ga.rad_blu, ga.mag_blu);
    ga.npass_blu = (int)h->rad_blu.size();
    ga.blu_chirp = h->d_blu_chirp;
    ga.blu_bhat = h->d_blu_bhat;
    ga.tw_blu = h->d_twg_blu;
  }
  ga.zp_full = (h->M > 1 && h->zp_full) ? 1 : 0;
  ga.zn = h->zn;
  if (ga.zp_full) {
    auto put_dft = [&](const fdoct_ctx::GenericDftPlan& p, GenericDft& d) {
      d.n = p.n;
      d.blu_m = p.blu_m;
      d.npass = (int)p.rad.size();
      put_plan(p.rad, d.rad, d.mag);
      d.tw = p.d_tw;
      d.chirp = p.d_chirp;
      d.bhat = p.d_bhat;
    };
    put_dft(h->gzf, ga.zf);
    put_dft(h->gzi, ga.zi);
  }
  ga.bandpass = h->bandpass ? 1 : 0;
  ga.inplace = h->generic_inplace ? 1 : 0;
  ga.radix16 = h->generic_radix16 ? 1 : 0;
  ga.rowwisenormalize = h->cfg.rowwisenormalize;
  ga.dcmask = h->cfg.dc_mask;
  ga.inv_A = (float)(1.0 / (double)A);
  ga.eps = chain_eps(h);
  ga.db_scale = kDbScale;
  ga.out_mag = c.k_mag;
  ga.out_db = c.k_db;
  {
    // the launch's row counter: one of 64 words used in turn, zeroed on the stream in front of the launch (no host
    // synchronisation; 64 launches of one handle are never in flight together)
    constexpr unsigned kGenTickets = 64;
    static const bool no_tickets = [] { const char* e = std::getenv("FDOCT_GENERIC_TICKETS"); return e && std::atoi(e) == 0; }();  // measurement: the static stride
    if (h->d_gen_tickets && !no_tickets) {   // (allocated with the kernel's tables: rebuild_generic_state)
      unsigned* t = h->d_gen_tickets + (h->gen_ticket_seq++ % kGenTickets);
      HIP_TRY(h, hipMemsetAsync(t, 0, sizeof(unsigned), c.st));
      ga.row_ticket = t;
    }
  }
  const size_t glds = generic_lds_bytes(h);
  int per_cu = (int)((160 * 1024 - 1024) / glds);
  if (per_cu > 6) per_cu = 6;  // generic_kernel is compiled for 6 waves per SIMD = 6 workgroups of 4 waves per CU
  if (per_cu < 1) per_cu = 1;
  long long ggrid = (long long)h->num_cu * per_cu;
  if (ggrid > c.out_rows) ggrid = c.out_rows;
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[1], c.st));
  HIP_TRY(h, launch_generic(ga, (int)ggrid, glds, c.st));
  return finish_launch(h, r, c, false);
}

#if defined(FDOCT_RUNTIME_ABLATE) || defined(FDOCT_CLOCKPROBE) || defined(FDOCT_FUSED_PROBE)
#define FDOCT_DEV_BUILD 1
// Measurement builds only (tools/ablate.sh, tools/mkvariant.sh probe): the stage-skipping mask and the in-kernel clock probes.
void dev_build_hooks(FusedArgs& a, hipStream_t st) {
#ifdef FDOCT_RUNTIME_ABLATE
  static const int ablate = [] { const char* ab = std::getenv("FDOCT_ABLATE"); return ab ? std::atoi(ab) : 0; }();
  a.ablate = ablate;
#endif
#ifdef FDOCT_FUSED_PROBE
  {  // share of a wave's cycles per phase of the row loop, first workgroups' waves, printed every 50 launches
    static unsigned long long* d_ph = nullptr;
    const size_t n = 4 * 16 * FUSED_PROBE_PHASES;
    if (!d_ph) {
      (void)hipMalloc(reinterpret_cast<void**>(&d_ph), n * 8);
      (void)hipMemset(d_ph, 0, n * 8);
    }
    a.phase_probe = d_ph;
    static int calls = 0;
    if (++calls % 50 == 0) {
      std::vector<unsigned long long> v(n);
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(v.data(), d_ph, n * 8, hipMemcpyDeviceToHost);
      static const char* names[10] = {"loads' tail + A2/A3", "window + slope + staging", "prefetch issue + gather", "FFT step 1 (registers)", "row swap",
                                      "twiddle + radix 4 + LDS writes", "LDS read-back", "step-5 twiddles + radix 16", "untangle + magnitude", "epilogue + stores"};
      double sum[10] = {}, tot = 0;
      int nw = 0;
      for (int w = 0; w < 64; w++) {
        if (!v[w * FUSED_PROBE_PHASES + 3]) continue;
        nw++;
        for (int i = 0; i < 10; i++) sum[i] += (double)v[w * FUSED_PROBE_PHASES + i];
      }
      for (int i = 0; i < 10; i++) tot += sum[i];
      std::fprintf(stderr, "[fused probe] %d waves, %.0f cycles per wave:", nw, nw ? tot / nw : 0.0);
      for (int i = 0; i < 10; i++) std::fprintf(stderr, " | %s %.1f%%", names[i], 100.0 * sum[i] / (tot > 0 ? tot : 1));
      std::fprintf(stderr, "\n");
    }
  }
#endif
#ifdef FDOCT_CLOCKPROBE
  static unsigned long long* d_probe = nullptr;
  const size_t pbytes = (32 + 1024) * 8;
  if (!d_probe) {
    (void)hipMalloc(reinterpret_cast<void**>(&d_probe), pbytes);
    (void)hipMemset(d_probe, 0, pbytes);
  }
  a.probe = d_probe;
  static int calls = 0;
  if (++calls % 64 == 0) {
    std::vector<unsigned long long> v(32 + 1024);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(v.data(), d_probe, pbytes, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[probe]");
    for (int w = 0; w < 16; w++)
      if (v[2 * w + 1]) std::fprintf(stderr, " w%d %.0fus@%.2fGHz", w, v[2 * w + 1] / 100.0, v[2 * w] / (v[2 * w + 1] * 10.0));
    unsigned long long t0 = ~0ull;
    std::vector<double> stv, en;
    for (int b = 0; b < 512; b++)
      if (v[32 + 2 * b]) t0 = std::min(t0, v[32 + 2 * b]);
    for (int b = 0; b < 512; b++)
      if (v[32 + 2 * b]) {
        stv.push_back((v[32 + 2 * b] - t0) / 100.0);
        en.push_back((v[32 + 2 * b + 1] - t0) / 100.0);
      }
    if (!stv.empty()) {
      std::sort(stv.begin(), stv.end());
      std::sort(en.begin(), en.end());
      std::fprintf(stderr, "\n[probe] %zu blocks: start max %.1f us; end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us", stv.size(),
                   stv.back(), en.front(), en[en.size() / 10], en[en.size() / 2], en[en.size() * 9 / 10], en.back());
    }
    std::fprintf(stderr, "\n");
  }
#else
  (void)st;
#endif
}
#endif

int launch_family_fused(fdoct_ctx* h, const Route& r, const Call& c) {
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  int rc;
  hipStream_t st = c.st;
  FusedArgs a{};
  a.frames = c.kframes;
  a.frames_lo = c.kframes_lo;
  a.pitch_bytes = (long long)r.kpitch;
  a.total_out_rows = c.out_rows;
  a.W = W;
  a.H = H;
  a.D = D;
  a.A = A;
  a.split = h->split;
  a.scratch_bytes = h->scratch_bytes;
  a.tw_count = h->tw_count;
  a.ib = h->d_ib;
  a.ib2d = h->d_ib2d_f;
  a.il = h->d_il;
  a.il2d = h->d_il2d_f;
  a.ilp = h->d_il_p;
  a.il16 = h->d_il16;
  a.il16_2d = h->d_il16_2d;
  a.yp = h->d_yp;
  a.yp_2d = h->yp.rows > 1;
  a.yd = h->d_yd;
  a.yd_2d = h->yd.rows > 1;
  a.win = h->d_win;
  a.g = h->d_g;
  a.gidx = h->d_gidx;
  a.tw = h->d_tw;
  a.utw = h->d_utw;
  a.phase = h->d_phase;
  a.minmax = r.need_minmax ? h->d_minmax : nullptr;
  a.rowwisenormalize = h->cfg.rowwisenormalize;
  a.dcmask = h->cfg.dc_mask;
  a.need_rc = (a.ib2d || a.yp_2d || a.yd_2d || a.minmax || a.frames_lo) ? 1 : 0;
  a.inv_A = (float)(1.0 / (double)A);
  a.eps = chain_eps(h);
  a.db_scale = kDbScale;
  a.out_mag = c.k_mag;
  a.out_db = c.k_db;
#ifdef FDOCT_DEV_BUILD
  dev_build_hooks(a, st);
#endif

  const FusedPlan& p = h->plan;
  // the unpredicated fast-path kernel applies to the plain acquisition configuration
  // (a full-frame background keeps the fast path on the row-swap plan: its resident registers prefetch the frame row)
  const bool fast_opts = fused_resident_consts(p.kind, true, A > 1, p.WCH, 0) && c.out_rows < 0x7fffffffLL && !h->staged;
  // (a full-frame background with the two-word reciprocal -- fdoct_set_precise_division -- runs on the any-option
  // kernel: the fast path's prefetch registers hold one word per sample)
  const bool bg_ok = h->yb.rows == 1 || (fast_opts && (!h->precise_div || fused_il_half(true, p.WCH)));
  const bool norm_ok = !a.minmax || fast_opts;  // whole-frame normalisation has a fast-path variant there too
  const bool lean = (r.kdt == FDOCT_K_U16 || r.kdt == FDOCT_K_U8) && W == 8 * p.T * p.WCH && bg_ok && !a.yp && !a.yd &&
                    (!a.rowwisenormalize || fast_opts) && norm_ok && !h->force_general;
  // launch geometry: as many waves per workgroup as LDS and the register budget allow
  const int rpw = 64 / p.T;
  a.lds_planes = fused_resident_consts(p.kind, lean, A > 1, p.WCH, 0) ? 0 : 1;
  // 1/background as two floats (reciprocal_words): always on the any-option kernel, by fdoct_set_precise_division on the fast path
  a.prec = (lean && !h->precise_div) ? 0 : (h->yb.rows == 1 ? 1 : 2);
  // (the averaging fast-path kernels that keep their planes in LDS are bound by its capacity: a fourth 4 W-byte plane would cost
  // C4 a wave per CU, so they read the low words from global memory instead)
  // (only the kernels with more than 32 samples per lane have that form -- a compile-time property, fused_il_global: the others
  // read the row's low words at its top from the LDS plane, resident constants or not -- fused_kernel's ILX)
  // (staged mode: the kernel that reads the samples is the resample stage, compiled WITHOUT averaging whatever A is -- it runs
  // over input A-scans -- so it takes its low words from the LDS plane like every non-averaging kernel)
  if (a.prec == 1 && fused_il_global(lean, A > 1 && !h->staged, p.WCH, p.T)) a.prec = 3;
  const size_t lds_const = const_lds_bytes(h, a.lds_planes != 0, a.prec == 1, fused_il_half(lean, p.WCH));
  const size_t lds_max = 160 * 1024 - 64;  // the kernel's static row-ticket counter lives in LDS too
  const int max_block = fused_max_block(h->NC, p.T, lean, p.kind);
  int max_waves = max_block / 64;
  // (a full-frame background with both words: every wave has a slot for the prefetched pattern row of its next A-scan)
  const size_t dma_per_wave = fused_il16_dma_bytes(lean && h->yb.rows > 1, a.prec == 2 && fused_il_half(lean, p.WCH), r.tro, 8 * p.T * p.WCH);
  const size_t per_wave = (size_t)h->scratch_bytes * rpw + dma_per_wave;
  int waves = (int)((lds_max - lds_const) / per_wave);
  if (waves > max_waves) waves = max_waves;
  if (h->block_override) {
    int w = h->block_override / 64;
    if (w >= 1 && w <= waves) waves = w;
  }
  if (waves < 1) return fail(h, FDOCT_ERR_UNSUPPORTED, "row does not fit in LDS");
  const size_t lds = lds_const + (size_t)waves * per_wave;
  const int blocks_per_cu = (int)(lds_max / lds) > 0 ? (int)(lds_max / lds) : 1;
  const int wave_cap = (max_block / 64) / waves;  // register budget: max_block threads per CU
  int bpc = blocks_per_cu < wave_cap ? blocks_per_cu : wave_cap;
  if (bpc < 1) bpc = 1;
  long long need = (c.out_rows + (long long)waves * rpw - 1) / ((long long)waves * rpw);
  long long grid = (long long)h->num_cu * bpc;
  if (h->grid_override > 0) grid = h->grid_override;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;

  size_t lds_launch = lds;
  int block_launch = waves * 64;
  if (r.tro) {
    if (!lean) return fail(h, FDOCT_ERR_DEVICE, "internal: fused transposed store chosen for a configuration off the fast path");
    // computing waves + the write-out wave; LDS: constants, one row buffer per computing wave, the ring of finished rows.
    // As many computing waves as the register budget allows, then the largest ring that fits (round 5: the store is bound by
    // how much of the next tile fits into the ring while a tile drains, so the kernel keeps its once-read tables out of LDS);
    // a wave is given up only where not even the smallest ring fits next to them.
    const size_t tro_const = tro_const_lds_bytes(h, r.kdt == FDOCT_K_U8 ? 1 : 2, a.minmax != nullptr);
    const int ww = fused_tro_writer_waves();
    int cw = max_waves - ww;
    if (h->block_override && h->block_override / 64 - ww >= 1 && h->block_override / 64 - ww < cw) cw = h->block_override / 64 - ww;
    static const unsigned ring_cap = [] { const char* e = std::getenv("FDOCT_TRO_RING"); return e ? (unsigned)std::atoi(e) : 0u; }();  // measurement: at most this many slots
    unsigned slots = 0;
    if (rpw == 4) {
      // whole groups of four waves, as many as registers and LDS allow; no ring (the rows wait in the waves' own buffers)
      constexpr int GW = fused_tro_group_waves();
      while (cw >= GW && tro_const + (size_t)cw * h->scratch_bytes * rpw > lds_max) cw--;
      cw = cw / GW * GW;
      if (cw < GW) return fail(h, FDOCT_ERR_DEVICE, "internal: no group of waves fits the LDS (transposed store, four rows per wave)");
    } else {
      for (; cw >= 1; cw--) {
        slots = fused_tro_ring_pick(lds_max - tro_const - (size_t)cw * h->scratch_bytes * rpw, D, rpw);
        if (slots) break;
      }
      if (cw < 1 || !slots) return fail(h, FDOCT_ERR_DEVICE, "internal: no LDS left for the transposed store's ring");
      if (ring_cap >= 20 && slots > ring_cap) slots = fused_tro_ring_pick((size_t)ring_cap * (size_t)(D + 4) * 4, D, rpw);
    }
    const size_t ring = (size_t)slots * (size_t)(D + 4) * 4;
    a.tr_ring = slots;
    block_launch = (cw + ww) * 64;
    lds_launch = tro_const + (size_t)cw * h->scratch_bytes * rpw + ring;
    const unsigned tile_rows = (unsigned)fused_tro_tile_rows(rpw);
    const unsigned tpf = (unsigned)((H + tile_rows - 1) / tile_rows);
    const long long tiles = (long long)c.G * tpf;
    grid = h->grid_override > 0 ? h->grid_override : h->num_cu;   // one workgroup per CU (the ring fills its LDS)
    if (grid > tiles) grid = tiles;
    if (!h->d_tro_fault) {
      // (coherent, mapped host memory; the kernels raise the word with a plain system-scope STORE of 1 -- a read-modify-write
      // atomic on host memory would need PCIe AtomicOps on the link and is dropped silently where they are missing: ADVICE r4)
      HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->d_tro_fault), sizeof(unsigned), hipHostMallocCoherent | hipHostMallocMapped));
      *h->d_tro_fault = 0u;
    }
    h->tro_used = true;
    a.tr_fault = h->d_tro_fault;
    a.tro = 1;
    a.tr_tpf = tpf;
    a.tr_tpf_magic = tpf > 1 ? (unsigned)((1ull << 32) / tpf) : 0xffffffffu;
    a.tr_total_tiles = (unsigned)tiles;
  }
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[1], st));
  if (h->staged) {
    if (!lean || r.kdt != FDOCT_K_U16)
      return fail(h, FDOCT_ERR_UNSUPPORTED, "staged mode is built for the plain u16 acquisition configuration only");
    // one k-linear row per INPUT A-scan between the stages: the resample stage runs over the in_rows input rows as they
    // lie (A = 1), the FFT stage gathers the A rows of an output A-scan
    if ((rc = dev_reserve(h, &h->ws_ylin, &h->ws_ylin_cap, (size_t)c.in_rows * h->NC * sizeof(float2)))) return rc;
    a.ylin = h->ws_ylin;
    h->ylin_rows = c.in_rows;
    {
      FusedArgs a1 = a;
      a1.stage = 1;
      a1.A = 1;
      a1.inv_A = 1.f;
      a1.total_out_rows = c.in_rows;
      long long need1 = (c.in_rows + (long long)waves * rpw - 1) / ((long long)waves * rpw);
      long long grid1 = h->grid_override > 0 ? h->grid_override : (long long)h->num_cu * bpc;
      if (grid1 > need1) grid1 = need1;
      HIP_TRY(h, launch_fused(p, a1, r.kdt, h->cplx, lean, (int)grid1, waves * 64, lds, st));
    }
    if (h->record_now) HIP_TRY(h, hipEventRecord(h->ev[4], st));
    a.stage = 2;
    HIP_TRY(h, launch_fused(p, a, r.kdt, h->cplx, lean, (int)grid, waves * 64, lds, st));
  } else {
    h->ylin_rows = 0;
    HIP_TRY(h, launch_fused(p, a, r.kdt, h->cplx, lean, (int)grid, block_launch, lds_launch, st));
  }
  return finish_launch(h, r, c, h->staged);
}

// Enqueue the whole path for device-resident frames.  d_out_* are row-major or
// transposed per `layout`.
int enqueue_one(fdoct_ctx* h, const void* d_frames, fdoct_dtype dtype, int nframes, size_t pitch_bytes,
                float* d_out_bscan, float* d_out_db, fdoct_layout layout) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!d_frames || nframes <= 0) return fail(h, FDOCT_ERR_INVALID, "no frames");
  if (nframes % h->A) return fail(h, FDOCT_ERR_INVALID, "nframes must be a multiple of averages");
  if (!h->yb.rows) return fail(h, FDOCT_ERR_STATE, "no background set (fdoct_set_background)");
  if (!d_out_bscan && !d_out_db) return fail(h, FDOCT_ERR_INVALID, "no output requested");
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  if (pitch_bytes == 0) pitch_bytes = es * h->W * h->fe_binx;
  if (pitch_bytes < es * h->W) return fail(h, FDOCT_ERR_INVALID, "pitch smaller than a row");
  DEVICE_SCOPE(h);
  int rc;
  Route r;
  if ((rc = choose_route(h, dtype, (uintptr_t)d_frames, pitch_bytes, (uintptr_t)d_out_bscan, (uintptr_t)d_out_db, layout, nframes, &r))) return rc;
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  Call c;
  c.st = h->stream;
  c.nframes = nframes;
  c.G = nframes / A;
  c.in_rows = (long long)nframes * H;
  c.out_rows = (long long)c.G * H;
  c.es = es;
  c.d_out_bscan = d_out_bscan;
  c.d_out_db = d_out_db;
  c.kframes = d_frames;
  hipStream_t st = c.st;
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[0], st));
  // ---- passes in front of the chain
  int kdt_now = kernel_dtype(dtype);
  size_t pitch_now = pitch_bytes;
  if (r.frontend) {  // raw camera frames: medianBlur + binning first (main:953-958); the caller's pitch describes the RAW rows
    void* fo = nullptr;
    size_t fp = 0;
    if ((rc = run_frontend(h, d_frames, kdt_now, nframes, W * h->fe_binx, H * h->fe_biny, pitch_bytes, h->fe_median, h->fe_binx, h->fe_biny, &fo, &fp)))
      return rc;
    c.kframes = fo;
    pitch_now = fp;
  }
  if (r.narrow_f64 && r.movavg) {  // doubles through smoothmovavg (main:987-991): the tap sums in double, then the two planes
    if ((rc = dev_reserve(h, &h->ws_mov, &h->ws_mov_cap, (size_t)c.in_rows * W * 4))) return rc;
    if ((rc = dev_reserve(h, &h->ws_mov_lo, &h->ws_mov_lo_cap, (size_t)c.in_rows * W * 4 + 32))) return rc;
    HIP_TRY(h, launch_movavg_f64(static_cast<const double*>(d_frames), (long long)(pitch_bytes / 8), W, c.in_rows, h->cfg.movavgn, h->ws_mov, h->ws_mov_lo, st));
    c.kframes = h->ws_mov;
    c.kframes_lo = h->ws_mov_lo;
    pitch_now = (size_t)W * 4;
    kdt_now = FDOCT_K_F32;
  } else if (r.narrow_f64) {  // data_y doubles (main:987): split once into two f32 planes on the device, x = hi + lo
    if ((rc = dev_reserve(h, &h->ws_f32, &h->ws_f32_cap, (size_t)c.in_rows * W * 4))) return rc;
    if ((rc = dev_reserve(h, &h->ws_f32_lo, &h->ws_f32_lo_cap, (size_t)c.in_rows * W * 4 + 32))) return rc;
    HIP_TRY(h, launch_f64_split(static_cast<const double*>(d_frames), (long long)(pitch_bytes / 8), h->ws_f32, h->ws_f32_lo, W, c.in_rows, st));
    c.kframes = h->ws_f32;
    c.kframes_lo = h->ws_f32_lo;
    pitch_now = (size_t)W * 4;
    kdt_now = FDOCT_K_F32;
  }
  if (r.movavg && r.mov_lo && !r.narrow_f64) {  // ... of float frames: sums in double, two planes
    if ((rc = dev_reserve(h, &h->ws_mov, &h->ws_mov_cap, (size_t)c.in_rows * W * 4))) return rc;
    if ((rc = dev_reserve(h, &h->ws_mov_lo, &h->ws_mov_lo_cap, (size_t)c.in_rows * W * 4 + 32))) return rc;
    HIP_TRY(h, launch_movavg_f32_wide(static_cast<const float*>(c.kframes), (long long)(pitch_now / 4), W, c.in_rows, h->cfg.movavgn, h->ws_mov, h->ws_mov_lo, st));
    c.kframes = h->ws_mov;
    c.kframes_lo = h->ws_mov_lo;
    pitch_now = (size_t)W * 4;
    kdt_now = FDOCT_K_F32;
  } else if (r.movavg && !r.narrow_f64) {  // smoothmovavg (main:990-991) runs before everything else, on the raw samples
    if ((rc = dev_reserve(h, &h->ws_mov, &h->ws_mov_cap, (size_t)c.in_rows * W * 4))) return rc;
    HIP_TRY(h, launch_movavg(c.kframes, kdt_now, (long long)pitch_now, W, c.in_rows, h->cfg.movavgn, h->ws_mov, st));
    c.kframes = h->ws_mov;
    pitch_now = (size_t)W * 4;
    kdt_now = FDOCT_K_F32;
  }
  if (kdt_now != r.kdt || pitch_now != r.kpitch) return fail(h, FDOCT_ERR_DEVICE, "internal: the route and the passes in front of the chain disagree");
  if (r.need_minmax) {
    // [nframes] results followed by the fast kernel's per-workgroup partials
    const size_t mm_elems = (size_t)nframes + (size_t)minmax_partial_count(nframes);
    if ((rc = dev_reserve(h, &h->d_minmax, &h->minmax_cap, mm_elems * sizeof(float2)))) return rc;
    HIP_TRY(h, launch_minmax(c.kframes, r.kdt, (long long)r.kpitch, W, H, nframes, h->d_yd, h->yd.rows > 1, h->d_minmax,
                             h->d_minmax + nframes, st));
  }
  // ---- where the chain writes: the caller's arrays, or the transpose pass's input
  c.k_mag = d_out_bscan;
  c.k_db = d_out_db;
  if (r.transpose_pass) {
    const size_t bytes = (size_t)c.out_rows * D * 4;
    if ((rc = dev_reserve(h, &h->ws_tr, &h->ws_tr_cap, bytes * 2))) return rc;
    if (d_out_bscan) c.k_mag = h->ws_tr;
    if (d_out_db) c.k_db = h->ws_tr + (size_t)c.out_rows * D;
  }
  switch (r.family) {
    case FDOCT_KERNEL_WAVE:
    case FDOCT_KERNEL_WAVE_JIT: return launch_family_wave(h, r, c);
    case FDOCT_KERNEL_LONG_ROWS: return launch_family_long_rows(h, r, c);
    case FDOCT_KERNEL_GENERIC: return launch_family_generic(h, r, c);
    default: return launch_family_fused(h, r, c);
  }
}

// The whole path for device-resident frames.  The reference's own layout (bscan is D x H, main:1220) is produced by the
// chain writing row-major B-scans into a library-owned intermediate and a transpose pass; a long batch is cut into chunks of
// whole B-scans whose intermediate (tr_chunk_bytes, reused by every chunk) is small enough to stay in the 256 MB Infinity
// Cache between the two kernels, so that per A-scan only the camera samples and the final image cross HBM.
int enqueue(fdoct_ctx* h, const void* d_frames, fdoct_dtype dtype, int nframes, size_t pitch_bytes,
            float* d_out_bscan, float* d_out_db, fdoct_layout layout) {
  if (!h) return FDOCT_ERR_INVALID;
  if (h->d_tro_fault && *static_cast<volatile unsigned*>(h->d_tro_fault)) {  // raised by an earlier asynchronous call
    *static_cast<volatile unsigned*>(h->d_tro_fault) = 0u;
    return fail(h, FDOCT_ERR_DEVICE, "transposed store: a wave of an earlier call timed out waiting for its tile buffer; that call's results are invalid");
  }
  h->rec_first = h->rec_last = true;
  if (layout != FDOCT_LAYOUT_TRANSPOSED_DxH || nframes <= 0 || (nframes % h->A) || !d_frames)
    return enqueue_one(h, d_frames, dtype, nframes, pitch_bytes, d_out_bscan, d_out_db, layout);
  const int G = nframes / h->A;
  const size_t per_group = (size_t)h->H * h->D * 4 * ((d_out_bscan ? 1 : 0) + (d_out_db ? 1 : 0));
  long long cg = per_group ? (long long)(h->tr_chunk_bytes / per_group) : G;
  if (cg < 1) cg = 1;
  if (G <= cg) return enqueue_one(h, d_frames, dtype, nframes, pitch_bytes, d_out_bscan, d_out_db, layout);
  if (h->yb.rows) {  // (without a background enqueue_one reports the error)
    int rc;
    if (h->dirty && (rc = rebuild_device_state(h))) return rc;
    if (fused_transposed_store_applies(h, dtype, d_frames, pitch_bytes, d_out_bscan, d_out_db, nframes))  // no intermediate at all
      return enqueue_one(h, d_frames, dtype, nframes, pitch_bytes, d_out_bscan, d_out_db, layout);
  }
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  const size_t pitch = pitch_bytes ? pitch_bytes : es * (size_t)h->W * h->fe_binx;
  const size_t frame_stride = pitch * (size_t)h->H * h->fe_biny;  // raw camera rows when a front end is set
  const size_t out_group = (size_t)h->H * h->D;
  uint64_t ascans = 0, bin = 0, bout = 0;
  for (long long g0 = 0; g0 < G; g0 += cg) {
    const int ng = (int)std::min<long long>(cg, G - g0);
    h->rec_first = g0 == 0;
    h->rec_last = g0 + ng >= G;
    const unsigned char* fr = static_cast<const unsigned char*>(d_frames) + (size_t)g0 * h->A * frame_stride;
    int rc = enqueue_one(h, fr, dtype, ng * h->A, pitch_bytes, d_out_bscan ? d_out_bscan + (size_t)g0 * out_group : nullptr,
                         d_out_db ? d_out_db + (size_t)g0 * out_group : nullptr, layout);
    if (rc) {
      h->rec_first = h->rec_last = true;
      return rc;
    }
    ascans += h->timing.ascans;
    bin += h->timing.bytes_in;
    bout += h->timing.bytes_out;
  }
  h->rec_first = h->rec_last = true;
  h->timing.ascans = ascans;
  h->timing.bytes_in = bin;
  h->timing.bytes_out = bout;
  return FDOCT_OK;
}

}  // namespace fdoct_impl

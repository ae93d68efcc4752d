// Training convolutions of the mid-size / small maps (fp16, raw output + BatchNorm statistics, and their dgrads) for gfx950 (CDNA4).
//
//   GEMM view as in conv_igemm.hip: M = N*Ho*Wo pixels, N = Cout, K = ntaps*Cin -- for M = 8192 .. 131072, N = 64 .. 512,
//   K = 128 .. 2304 (the 32x64 / 16x32 stages at batch 16 and the 3x3 layers of the segmentation head).  conv_igemm ran these at
//   0.06-0.14 of their per-layer roofline (profiles/r3j_conv_layers.md): one 4-wave workgroup per CU, ONE wave per SIMD, a 64-byte K
//   step staged global -> registers -> LDS with ~110 address instructions, 16 MFMAs and a barrier per step -- a serial chain per SIMD.
//
//   This kernel:
//     * 128-byte K steps (64 halves) staged with LDS-DMA (`global_load_lds_dwordx4`: no staging registers, no ds_write pass) into a
//       3- or 4-stage ring; the loads of step s+2 / s+3 are issued between the MFMA groups of step s, waits are COUNTED (`s_waitcnt
//       vmcnt(L)`: the younger stages stay in flight across the barrier), one raw `s_barrier` per step;
//     * 8 waves per 128 x 128 tile (two per SIMD: one wave's LDS reads sit under the other's MFMAs), 16 MFMAs per wave and step;
//     * the per-row addresses (pixel base offset, a bit mask of the taps that fall inside the image) are computed once per tile; a
//       K step costs two scalar table reads, one select and one add per load;
//     * LDS rows are 128 B, the 16-byte segment XOR-ed with (row >> 1) & 7: conflict-free for the ds_read_b128 lane groups of the
//       MFMA fragment pattern (brute-forced over all bases); the LDS-DMA image is lane-linear, so the swizzle is applied to the SOURCE
//       address (lane l of a 1 KB piece fetches logical segment (l & 7) ^ swizzle of row l >> 3);
//     * D^T = W . X^T over the row-permuted weight tile (panel_chan, myolo_dev.h): a lane owns 8 consecutive channels of one pixel ->
//       16-byte NHWC stores straight from the accumulators, BatchNorm statistics reduced over the 16 pixel lanes by shuffles.
//
// Replaces: nn.Conv2d in training mode (reference models/common.py:34-46: the conv of Conv / Bottleneck / C3 / SPP / PSP head) and its
// autograd dgrad.  Epilogues: raw store (+ statistics) (+ residual) (+ accumulate); BNS: + the BatchNorm-backward sums of the layer below
// (myolo_conv_desc.bnb); EPI: folded BatchNorm scale / shift + activation (eval, common.py:45-46); BNA: myolo_conv_dgrad_bn -- the layer's own
// BatchNorm-backward apply pass in the operand path of its 1x1 dgrad.  fp32, Detect's permuted output and odd channel counts stay on conv_igemm.
#include "myolo_dev.h"
#include <string.h>
#include <stdlib.h>

namespace mid {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

struct MidK {
  const char* x; const char* w; char* y; const char* res; float* stats;
  const float* scale; const float* shift; int act;      // EPI: y = act(conv * scale[c] + shift[c]) (eval: folded BatchNorm / bias + activation)
  int x_sn, x_sh, x_sw;            // bytes
  int y_sn, y_sh, y_sw;
  int r_sn, r_sh, r_sw;
  int Hi, Wi, Wo, HWo, M, Cout;
  int stride, ntaps, kchunks, nsteps;
  int wrow_bytes;                  // bytes between two output-channel rows of the packed weights (wtaps * cin_pad * 2)
  int cin_b;                       // bytes of one input pixel's channels (x.c * 2): below kchunks * 128 for yolov5m's 48 / 96-channel layers, whose
                                   // last K step is ragged -- the lanes whose 16-byte segment lies past it fetch the zero page (round 6)
  int ntile_p, tiles_per_xcd, accumulate;
  // BNA: the BatchNorm-backward APPLY pass of the layer this 1x1 dgrad belongs to, in its operand path: x is the gradient w.r.t. the
  // layer's activation output (gout), `by` its saved raw conv output; dy = sc*dz + cb*y + cd (dz = gout * act'(y*sc + sh)) is formed in LDS
  // once per 16-byte slot by the thread that loaded it, feeds the MFMAs and is stored to `bdy` (the weight gradient's operand) by the
  // workgroups of N tile 0; workgroup (0, 0) adds dgamma / dbeta
  const char* by; char* bdy; int by_sn, by_sh, by_sw, dy_sn, dy_sh, dy_sw;
  const float* b_saved; const float* b_gamma; const float* b_beta; const float* b_dsum; float* b_dgamma; float* b_dbeta;
  int b_act, b_K; float b_rM;      // channels of the layer (= this conv's K), 1 / pixels
  // FUS (round 6, myolo_conv_bn_act): training-mode Conv + BatchNorm + activation in one launch.  Every workgroup owns ONE tile and keeps its
  // accumulators across a device-wide barrier: statistics -> atomics -> barrier -> totals -> scale / shift -> y (raw) and out (activation)
  char* f_out; const char* f_res; int fo_sn, fo_sh, fo_sw, fr_sn, fr_sh, fr_sw;
  const float* f_gamma; const float* f_beta; const float* f_gamma2; const float* f_beta2;
  float* f_rm; float* f_rv; float* f_rm2; float* f_rv2; int64_t* f_nbt; int64_t* f_nbt2; float* f_saved;
  float f_eps, f_mom; int f_act, f_cs, f_world; unsigned int* f_bar;
  BnbArgs bnb;                     // BNS: BatchNorm-backward statistics of the layer(s) whose output gradient this launch completes
  int dbg;                         // profiling only: 1 no steady-state loads, 2 no fragment reads, 4 no MFMAs, 8 no stores
  int tap_dy[MYOLO_MAX_TAPS], tap_dx[MYOLO_MAX_TAPS];
  int tap_xoff[MYOLO_MAX_TAPS];    // (dy * x_sh + dx * x_sw) bytes
  int tap_woff[MYOLO_MAX_TAPS];    // tap_w[t] * cin_pad * 2 bytes
};

// sum over the 16 lanes of a DPP row, valid in lane 15 of the row: four v_add_f32 with a row_shr operand (no LDS crossbar permutes)
__device__ __forceinline__ float row_sum16(float v) {
#define MID_SHR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + (n), 0xf, 0xf, true))
  v += MID_SHR(1); v += MID_SHR(2); v += MID_SHR(4); v += MID_SHR(8);
#undef MID_SHR
  return v;
}

// BM x BN output tile (pixels x channels), WP x WC waves (each (BM/WP) pixels x (BN/WC) channels), NST LDS stages
// BNS: the stored gradient completes gout of a BatchNorm layer -> its backward sums (myolo_conv_desc.bnb; conv_igemm.hip has the same fold):
//   dsum0 += dz, dsum1 += dz * xhat with dz = gout * act'(bn(y)) on the final, storage-rounded values, per channel
// EPI (exclusive with BNS): per-channel scale / shift and activation ahead of the residual add (the eval epilogue of conv_igemm.hip)
template <int BM, int BN, int WP, int WC, int NST, bool DBG, bool BNS = false, bool EPI = false, bool BNA = false, bool FUS = false>
__global__ __launch_bounds__(64 * WP * WC) void conv_mid_kernel(const MidK p) {
  static_assert(!(BNS && EPI) && !(BNA && EPI), "one epilogue table");
  static_assert(!FUS || (!BNS && !EPI && !BNA && !DBG), "the fused forward is the plain statistics kernel plus its tail");
  constexpr int NT = 64 * WP * WC;
  constexpr int PW = BM / WP, CW = BN / WC;       // wave tile
  constexpr int PF = PW / 16, CF = CW / 16;       // 16 x 16 fragments per wave
  static_assert(CF % 2 == 0, "channel fragments are stored in pairs (8 consecutive channels per lane)");
  constexpr int RPI = NT / 8;                     // rows per load instruction of the workgroup (8 lanes per 128-byte row)
  constexpr int XR = BM / RPI, WR = BN / RPI;     // loads per thread and stage
  static_assert(XR >= 1 && WR >= 1 && BM % RPI == 0 && BN % RPI == 0, "tile rows must divide over the loader lanes");
  constexpr int LPS = XR * (BNA ? 2 : 1) + WR;    // BNA: a second pixel tile (the layer's raw conv output) per stage, behind the weights
  constexpr int STAGE = (BM * (BNA ? 2 : 1) + BN) * 128;
  constexpr int MAXK = 512;                       // BNA: channels of the constant table
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave / WC, wc = wave % WC;
  const int lq = lane >> 4, l15 = lane & 15;
  const int tn = blockIdx.y;
  const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3, bstride = gridDim.x >> 3;

  // loader role: row (tid >> 3) (+ j * RPI), physical segment tid & 7 <- logical segment lsg
  const int lrow = tid >> 3;
  const int lsg = (tid & 7) ^ ((lrow >> 1) & 7);
  int wbase[WR];
#pragma unroll
  for (int j = 0; j < WR; ++j) wbase[j] = (tn * BN + panel_chan(j * RPI + lrow)) * p.wrow_bytes + lsg * 16;
  const unsigned klast = (unsigned)((p.kchunks - 1) * 128 + lsg * 16 < p.cin_b);      // this lane's segment of the LAST K chunk holds channels

  // fragment reads: row l15 of a 16-row fragment, K segment kk*4 + lq, swizzled
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)smem;
  unsigned foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = l15 * 128 + (((kk * 4 + lq) ^ ((l15 >> 1) & 7)) << 4);

  // lane t holds tap t's (dy, dx): ONE vector read of the kernel-argument table
  const int v_dy = p.tap_dy[lane < MYOLO_MAX_TAPS ? lane : 0], v_dx = p.tap_dx[lane < MYOLO_MAX_TAPS ? lane : 0];

  float st_s[CF / 2][8], st_q[CF / 2][8];
#pragma unroll
  for (int q = 0; q < CF / 2; ++q)
#pragma unroll
    for (int i = 0; i < 8; ++i) { st_s[q][i] = 0.f; st_q[q][i] = 0.f; }

  int sgi = -1;                                     // BNS: the segment this N tile belongs to (host: segments are BN-aligned)
  float* sBN = reinterpret_cast<float*>(smem + NST * STAGE);      // [4][BN] mean, invstd, gamma*invstd, beta - mean*gamma*invstd
  if (BNS) {
    for (int i = 0; i < p.bnb.n; ++i)
      if (tn * BN >= p.bnb.seg[i].c0 && tn * BN < p.bnb.seg[i].c1) sgi = i;
    if (sgi >= 0) {
      const BnbSeg& sg = p.bnb.seg[sgi];
      const int Cs = sg.c1 - sg.c0;
      for (int c = tid; c < BN; c += NT) {
        const int ci = tn * BN + c - sg.c0;
        const bool in = ci < Cs;
        const float mean = in ? sg.saved[ci] : 0.f, istd = in ? sg.saved[Cs + ci] : 0.f;
        const float sc = in ? sg.gamma[ci] * istd : 0.f;
        sBN[c] = mean; sBN[BN + c] = istd; sBN[2 * BN + c] = sc; sBN[3 * BN + c] = in ? sg.beta[ci] - mean * sc : 0.f;
      }
    }
    __syncthreads();
  }
  if (EPI) {                                        // the same LDS area: [2][BN] scale, shift of this N tile
    for (int c = tid; c < BN; c += NT) {
      const int cg = tn * BN + c;
      sBN[c] = (p.scale && cg < p.Cout) ? p.scale[cg] : 1.f;
      sBN[BN + c] = (p.shift && cg < p.Cout) ? p.shift[cg] : 0.f;
    }
    __syncthreads();
  }

  float* sBA = sBN + (BNS ? 4 * BN : 0);            // BNA: [4][b_K] sc, sh, cb, cd of the layer being differentiated (bn_act.hip's apply constants)
  if (BNA) {
    const int K = p.b_K;
    for (int c = tid; c < K; c += NT) {
      const float mean = p.b_saved[c], istd = p.b_saved[K + c];
      const float sc = p.b_gamma[c] * istd;
      float d0 = 0.f, d1 = 0.f;
#pragma unroll 8
      for (int k = 0; k < MYOLO_STAT_COPIES; ++k) { d0 += p.b_dsum[k * 2 * K + c]; d1 += p.b_dsum[k * 2 * K + K + c]; }
      const float cb = -sc * (d1 * p.b_rM) * istd;
      sBA[c] = sc; sBA[K + c] = p.b_beta[c] - mean * sc; sBA[2 * K + c] = cb; sBA[3 * K + c] = -sc * (d0 * p.b_rM) - cb * mean;
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (p.b_dgamma) p.b_dgamma[c] += d1;
        if (p.b_dbeta) p.b_dbeta[c] += d0;
      }
    }
    __syncthreads();
  }

  bool fus_done = false;
  for (int tslot = bslot; tslot < p.tiles_per_xcd; tslot += bstride) {
    const int tp = xcd * p.tiles_per_xcd + tslot;
    if (tp >= p.ntile_p) break;
    const int m0 = tp * BM;

    int xbase[XR]; unsigned xmask[XR];
    int ybase[XR], dybase[XR];                     // BNA: the row's pixel in the raw conv output / dy tensors
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int m = m0 + j * RPI + lrow;
      const bool valid = m < p.M;
      const int mm = valid ? m : 0;
      const int n = mm / p.HWo; const int rem = mm - n * p.HWo;
      const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
      const int iy0 = oy * p.stride, ix0 = ox * p.stride;
      xbase[j] = n * p.x_sn + iy0 * p.x_sh + ix0 * p.x_sw + lsg * 16;
      if (BNA) {
        ybase[j] = n * p.by_sn + oy * p.by_sh + ox * p.by_sw + lsg * 16;
        dybase[j] = n * p.dy_sn + oy * p.dy_sh + ox * p.dy_sw + lsg * 16;
      }
      unsigned mk = 0;
      for (int t = 0; t < p.ntaps; ++t) {          // (tap offsets by readlane: a scalar table read per tap and row was a chain of ~0.25 us loads)
        const int iy = iy0 + __builtin_amdgcn_readlane(v_dy, t), ix = ix0 + __builtin_amdgcn_readlane(v_dx, t);
        mk |= (unsigned)(valid && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) << t;
      }
      xmask[j] = mk;
    }

    f4_t acc[CF][PF];
#pragma unroll
    for (int c = 0; c < CF; ++c)
#pragma unroll
      for (int q = 0; q < PF; ++q) acc[c][q] = f4_t{0.f, 0.f, 0.f, 0.f};

    int i_tap = 0, i_kc = 0;                       // issue cursor
    // table entries of the NEXT issue: the scalar reads run a step ahead and are first touched behind the next barrier (a use inside
    // this step would make hipcc wait lgkmcnt(0) in the middle of the fragment reads)
    int n_xt = p.tap_xoff[0], n_wt = p.tap_woff[0];
    const char* src[LPS];
    auto addresses = [&]() {                       // source addresses of the next stage's LPS pieces (+ cursor advance)
      const int t = i_tap;
      const int xo = n_xt + i_kc * 128, wo = n_wt + i_kc * 128;
      const unsigned km = (i_kc == p.kchunks - 1) ? klast : 1u;      // (uniform compare: a ragged last chunk costs one AND per load)
#pragma unroll
      for (int j = 0; j < XR; ++j)                 // per-lane select: taps outside the image / rows past M / channels past the tensor fetch the zero page; the load itself is unconditional
        src[j] = ((xmask[j] >> t) & km) ? p.x + (unsigned)(xbase[j] + xo) : zero_page();
#pragma unroll
      for (int j = 0; j < WR; ++j) src[XR + j] = p.w + (unsigned)(wbase[j] + wo);
      if (BNA) {
#pragma unroll
        for (int j = 0; j < XR; ++j) src[XR + WR + j] = ((xmask[j] >> t) & 1u) ? p.by + (unsigned)(ybase[j] + i_kc * 128) : zero_page();
      }
      if (++i_kc == p.kchunks) { i_kc = 0; ++i_tap; }
      const int tn_ = i_tap < p.ntaps ? i_tap : 0;
      n_xt = p.tap_xoff[tn_]; n_wt = p.tap_woff[tn_];
    };
    auto piece = [&](int i, int buf) {             // one 1 KB LDS-DMA piece of this wave: 8 rows x 128 B, lane-linear in LDS
      char* dst = smem + buf * STAGE + wave * 1024 +
                  (i < XR ? i * RPI * 128 : (i < XR + WR ? BM * 128 + (i - XR) * RPI * 128 : (BM + BN) * 128 + (i - XR - WR) * RPI * 128));
      __builtin_amdgcn_global_load_lds((gptr_t*)src[i], (lptr_t*)dst, 16, 0, 0);
    };
    // One K step.  The fragment reads and their waits are inline asm: hipcc's waitcnt pass makes every ds_read it can see wait for ALL
    // outstanding LDS-DMA (vmcnt(0): it cannot tell the stage being read from the stages in flight), which would serialise the ring.
    // The "+v" operands behind a wait tie the fragments to it, so the MFMAs that consume them cannot be scheduled above it.
    // The LDS-DMA pieces of stage s + NST-1 are issued BETWEEN the MFMA groups (an LDS-DMA issue holds its wave for 60-180 cycles,
    // MI355X_MICROARCH.md: issued in a block ahead of the reads -- both waves of a SIMD leave the barrier together -- the matrix pipe
    // idled through it; scripts/conv_train_ubench.py PROBE=mid_dbg: loads alone 4.9 us, MFMAs alone 5.4, together 11.5 of an 18-step tile).
    // BNA: this thread's own 16-byte slots of stage `buf` (the ones its LDS-DMA pieces filled: no barrier needed in front), K chunk kc:
    // gout and y -> dy in place of gout; all LDS traffic in inline asm like the fragment reads (hipcc would wait vmcnt(0) before each)
    auto transform = [&](int buf, int kc) {
      const unsigned ga = lds0 + buf * STAGE + tid * 16, ya = ga + (BM + BN) * 128;
      const unsigned ca = lds0 + NST * STAGE + (BNS ? 4 * BN * 4 : 0) + (kc * 64 + lsg * 8) * 4;
      const unsigned Kb = (unsigned)p.b_K * 4;
      u32x4_t cv[8], gv[XR], yv[XR];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(cv[2 * a]) : "v"(ca + a * Kb) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(cv[2 * a + 1]) : "v"(ca + a * Kb) : "memory");
      }
#pragma unroll
      for (int j = 0; j < XR; ++j) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(gv[j]) : "v"(ga), "n"(j * RPI * 128) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yv[j]) : "v"(ya), "n"(j * RPI * 128) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int a = 0; a < 8; ++a) asm volatile("" : "+v"(cv[a]));
#pragma unroll
      for (int j = 0; j < XR; ++j) { asm volatile("" : "+v"(gv[j])); asm volatile("" : "+v"(yv[j])); }
      const float* sc = reinterpret_cast<const float*>(&cv[0]);      // [8] each: sc, sh, cb, cd
      const float* sh = reinterpret_cast<const float*>(&cv[2]);
      const float* cb = reinterpret_cast<const float*>(&cv[4]);
      const float* cd = reinterpret_cast<const float*>(&cv[6]);
#pragma unroll
      for (int j = 0; j < XR; ++j) {
        float fg[8], fy[8], o[8];
        Vec<half_t>::unpack(uint4{gv[j].x, gv[j].y, gv[j].z, gv[j].w}, fg);
        Vec<half_t>::unpack(uint4{yv[j].x, yv[j].y, yv[j].z, yv[j].w}, fy);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float dz = fg[i] * silu_grad_f(fmaf(fy[i], sc[i], sh[i]));       // (host: SiLU layers only -- no per-element activation switch)
          o[i] = fmaf(sc[i], dz, fmaf(cb[i], fy[i], cd[i]));
        }
        if (!(xmask[j] & 1u)) {                     // rows past M: exact zeros in the MFMA operand, as in the two-launch form (the
#pragma unroll                                      // constant term cd would otherwise reach any epilogue that reduces over unmasked rows)
          for (int i = 0; i < 8; ++i) o[i] = 0.f;
        }
        const u32x4_t ov = pack_h8(o);
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ga), "v"(ov), "n"(j * RPI * 128) : "memory");
        if (tn == 0 && (xmask[j] & 1u)) stg16(p.bdy + (unsigned)(dybase[j] + kc * 128), uint4{ov.x, ov.y, ov.z, ov.w});
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the dy slots are in LDS before the barrier lets the fragment reads go
    };
    constexpr int G = 2 * CF;                      // MFMA groups (PF MFMAs each) per step
    auto step = [&](int buf, int nb, const bool do_issue) {
      const unsigned ax = lds0 + buf * STAGE + (wp * PW) * 128, aw = lds0 + buf * STAGE + BM * 128 + (wc * CW) * 128;
      u32x4_t wf[2][CF], xf[2][PF];
      if (DBG && (p.dbg & 2)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int c = 0; c < CF; ++c) wf[kk][c] = u32x4_t{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
          for (int q = 0; q < PF; ++q) xf[kk][q] = u32x4_t{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int c = 0; c < CF; ++c) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kk][c]) : "v"(aw + foff[kk]), "n"(c * 2048) : "memory");
#pragma unroll
          for (int q = 0; q < PF; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[kk][q]) : "v"(ax + foff[kk]), "n"(q * 2048) : "memory");
        }
      }
      const bool loads = do_issue && !(DBG && (p.dbg & 1));
      if (loads) addresses();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        __builtin_amdgcn_sched_barrier(0);         // (keeps the address arithmetic / the first half's MFMAs above the wait)
        if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CF + PF) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < CF; ++c) asm volatile("" : "+v"(wf[kk][c]));
#pragma unroll
        for (int q = 0; q < PF; ++q) asm volatile("" : "+v"(xf[kk][q]));
#pragma unroll
        for (int c = 0; c < CF; ++c) {
          if (!(DBG && (p.dbg & 4))) {
#pragma unroll
            for (int q = 0; q < PF; ++q)
              acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&wf[kk][c]), *reinterpret_cast<const h8_t*>(&xf[kk][q]),
                                                                 acc[c][q], 0, 0, 0);
          }
          const int g = kk * CF + c;
#pragma unroll
          for (int i = 0; i < LPS; ++i)
            if ((i * G) / LPS == g && loads) {
              __builtin_amdgcn_sched_barrier(0);
              piece(i, nb);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
    };

    // ring of NST stages: the loads of step s + NST-1 are issued during the MFMAs of step s.  A wait leaves the younger stages in
    // flight; the barrier behind it says that everybody's loads of stage s have landed AND that every wave is done reading the buffer of
    // step s-1, which this step's loads overwrite.  (nsteps >= NST-1: host)
    const int inflight = p.nsteps < NST - 1 ? p.nsteps : NST - 1;      // stages the prologue issues (short K loops: fewer than the ring holds)
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
      if (j < inflight) {
        addresses();
#pragma unroll
        for (int i = 0; i < LPS; ++i) piece(i, j);
      }
    int buf = 0, kcur = 0;
    const int steady = p.nsteps - (NST - 1);
    for (int s = 0; s < steady; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPS) : "memory");
      if (BNA) transform(buf, kcur++);
      __builtin_amdgcn_s_barrier();
      int nb = buf + NST - 1; nb = nb >= NST ? nb - NST : nb;
      step(buf, nb, true);
      buf = buf + 1 == NST ? 0 : buf + 1;
    }
#pragma unroll
    for (int r = NST - 2; r >= 0; --r) {           // drain: no more issues, r stages stay in flight
      if (r >= inflight) continue;                 // (uniform: a K loop shorter than the ring)
      if (r == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      else if (r == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
      if (BNA) transform(buf, kcur++);
      __builtin_amdgcn_s_barrier();
      step(buf, 0, false);
      buf = buf + 1 == NST ? 0 : buf + 1;
    }
    __builtin_amdgcn_s_barrier();                  // (the next tile's prologue overwrites buffers the slowest wave may still be reading)

    if (FUS) {
      // ---- fused forward tail (this workgroup's ONLY tile: host) ----
      // (1) per-channel sum / sum of squares of the raw fp32 accumulators (rows past M hold zeros: their operand rows were the zero page)
#pragma unroll
      for (int q = 0; q < PF; ++q)
#pragma unroll
        for (int h = 0; h < CF / 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = acc[2 * h][q][r], b = acc[2 * h + 1][q][r];
            st_s[h][r] += a; st_q[h][r] += a * a; st_s[h][4 + r] += b; st_q[h][4 + r] += b * b;
          }
      float* red = reinterpret_cast<float*>(smem);       // [WP][2][BN] (the ring is idle: the tile ended with a barrier)
#pragma unroll
      for (int h = 0; h < CF / 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float s_ = row_sum16(st_s[h][i]), q_ = row_sum16(st_q[h][i]);
          if (l15 == 15) {
            const int cl = wc * CW + 32 * h + 8 * lq + i;
            red[(wp * 2) * BN + cl] = s_; red[(wp * 2 + 1) * BN + cl] = q_;
          }
        }
      __syncthreads();
      float keep = 0.f;
      for (int t = tid; t < 2 * BN; t += NT) {
        const int which = t / BN, cl = t - which * BN;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < WP; ++k) a += red[(k * 2 + which) * BN + cl];
        const int c = tn * BN + cl;
        // RETURNING atomics: the add has been performed when the value is back (the barrier's arrival must not overtake it)
        if (c < p.Cout) keep += atomicAdd(p.stats + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * p.Cout + which * p.Cout + c, a);
      }
      asm volatile("" ::"v"(keep));
      // (2) every tile of the layer has added its sums
      grid_barrier_xcd(p.f_bar, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
      // (3) totals of this N tile's channels (all copies in flight, sc1 loads: written by other CUs' atomics) -> scale / shift
      float* tot = sBN + 2 * BN;                       // [2][BN]
      for (int t = tid; t < 2 * BN; t += NT) {
        const int which = t / BN, cl = t - which * BN;
        const int c = tn * BN + cl;
        const unsigned voff = (unsigned)(which * p.Cout + (c < p.Cout ? c : 0)) * 4u;
        float v[MYOLO_STAT_COPIES];
#pragma unroll
        for (int k = 0; k < MYOLO_STAT_COPIES; ++k) {
          const float* base = p.stats + (size_t)k * 2 * p.Cout;
          asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1" : "=&v"(v[k]) : "v"(voff), "s"(base) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < MYOLO_STAT_COPIES; ++k) asm volatile("" : "+v"(v[k]));
        float d4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < MYOLO_STAT_COPIES; ++k) d4[k & 3] += v[k];
        tot[t] = (d4[0] + d4[1]) + (d4[2] + d4[3]);      // (bn_act_fwd forms the same four chains; it adds them in double)
      }
      __syncthreads();
      for (int cl = tid; cl < BN; cl += NT) {
        const int c = tn * BN + cl;
        float sc = 0.f, sh = 0.f;
        if (c < p.Cout) {
          const int64_t Mg = (int64_t)p.M * p.f_world;
          const double meand = (double)tot[cl] / (double)Mg;
          const double vard = (double)tot[BN + cl] / (double)Mg - meand * meand;
          const float mean = (float)meand, var = vard > 0.0 ? (float)vard : 0.f;
          const float invstd = rsqrtf(var + p.f_eps);
          const bool lo = c < p.f_cs;
          const int cc = lo ? c : c - p.f_cs;
          sc = (lo ? p.f_gamma : p.f_gamma2)[cc] * invstd;
          sh = (lo ? p.f_beta : p.f_beta2)[cc] - mean * sc;
          if (blockIdx.x == 0) {                         // one workgroup per N tile publishes the layer's statistics
            if (p.f_saved) { p.f_saved[c] = mean; p.f_saved[p.Cout + c] = invstd; }
            float* rmp = lo ? p.f_rm : p.f_rm2; float* rvp = lo ? p.f_rv : p.f_rv2;
            if (rmp) {
              rmp[cc] = (1.f - p.f_mom) * rmp[cc] + p.f_mom * mean;
              const float unb = Mg > 1 ? var * (float)Mg / (float)(Mg - 1) : var;
              rvp[cc] = (1.f - p.f_mom) * rvp[cc] + p.f_mom * unb;
            }
          }
        }
        sBN[cl] = sc; sBN[BN + cl] = sh;
      }
      if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && p.f_nbt) *p.f_nbt += 1;
      if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 1 && p.f_nbt2 && p.f_cs < p.Cout) *p.f_nbt2 += 1;
      __syncthreads();
      // (4) raw output and activation from the same accumulators: the normalisation reads the STORED (fp16-rounded) value, as bn_act_fwd does
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int m = m0 + wp * PW + q * 16 + l15;
        const bool mvalid = m < p.M;
        const int mm = mvalid ? m : 0;
        const int n = mm / p.HWo; const int rem = mm - n * p.HWo;
        const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
        const unsigned yoff = (unsigned)(n * p.y_sn + oy * p.y_sh + ox * p.y_sw);
        const unsigned ooff = (unsigned)(n * p.fo_sn + oy * p.fo_sh + ox * p.fo_sw);
        const unsigned roff = (unsigned)(n * p.fr_sn + oy * p.fr_sh + ox * p.fr_sw);
        uint4 rv[CF / 2];
#pragma unroll
        for (int h = 0; h < CF / 2; ++h) {
          const int c0 = tn * BN + wc * CW + 32 * h + 8 * lq;
          if (p.f_res) rv[h] = ldg16((mvalid && c0 < p.Cout) ? p.f_res + roff + c0 * 2 : zero_page());
        }
#pragma unroll
        for (int h = 0; h < CF / 2; ++h) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[2 * h][q][r]; v[4 + r] = acc[2 * h + 1][q][r]; }
          const int c0 = tn * BN + wc * CW + 32 * h + 8 * lq;
          const int cl = wc * CW + 32 * h + 8 * lq;
          if (mvalid && c0 < p.Cout) {
            const u32x4_t y16 = pack_h8(v);
            stg16(p.y + yoff + c0 * 2, uint4{y16.x, y16.y, y16.z, y16.w});
            float z[8];
            Vec<half_t>::unpack(uint4{y16.x, y16.y, y16.z, y16.w}, z);
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = act_f(fmaf(z[i], sBN[cl + i], sBN[BN + cl + i]), p.f_act);
            if (p.f_res) add_h8(z, u32x4_t{rv[h].x, rv[h].y, rv[h].z, rv[h].w});
            const u32x4_t o = pack_h8(z);
            stg16(p.f_out + ooff + c0 * 2, uint4{o.x, o.y, o.z, o.w});
          }
        }
      }
      fus_done = true;
    } else {
    // ---- epilogue: lane = pixel l15 of fragment q, channels 32*h + 8*lq .. +7 of the wave's channel range (h = fragment pair) ----
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int m = m0 + wp * PW + q * 16 + l15;
      const bool mvalid = m < p.M;
      const int mm = mvalid ? m : 0;
      const int n = mm / p.HWo; const int rem = mm - n * p.HWo;
      const int oy = rem / p.Wo; const int ox = rem - oy * p.Wo;
      const unsigned yoff = (unsigned)(n * p.y_sn + oy * p.y_sh + ox * p.y_sw);
      const unsigned roff = (unsigned)(n * p.r_sn + oy * p.r_sh + ox * p.r_sw);
      uint4 yv[CF / 2];                            // BNS: the layer's raw conv output at this pixel (loads issued ahead of the stores)
      if (BNS) {
        if (sgi >= 0) {
          const BnbSeg& sg = p.bnb.seg[sgi];
          const unsigned boff = (unsigned)(n * (int)sg.y_sn + oy * (int)sg.y_sh + ox * (int)sg.y_sw) * 2u;
#pragma unroll
          for (int h = 0; h < CF / 2; ++h) {
            const int c0 = tn * BN + wc * CW + 32 * h + 8 * lq;
            yv[h] = ldg16((mvalid && c0 < sg.c1) ? sg.y + boff + (c0 - sg.c0) * 2 : zero_page());
          }
        }
      }
      // residual / accumulate operands of all channel groups first: one L2 round trip per pixel fragment, not one per 16-byte store
      // (hipcc waits vmcnt(0) at the first use of each load)
      uint4 rv[CF / 2], av[CF / 2];
#pragma unroll
      for (int h = 0; h < CF / 2; ++h) {
        const int c0 = tn * BN + wc * CW + 32 * h + 8 * lq;
        const bool ok = mvalid && c0 < p.Cout;
        if (p.res) rv[h] = ldg16(ok ? p.res + roff + c0 * 2 : zero_page());
        if (p.accumulate) av[h] = ldg16(ok ? p.y + yoff + c0 * 2 : zero_page());
      }
#pragma unroll
      for (int h = 0; h < CF / 2; ++h) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[2 * h][q][r]; v[4 + r] = acc[2 * h + 1][q][r]; }
        if (!BNS) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { st_s[h][i] += v[i]; st_q[h][i] += v[i] * v[i]; }
        }
        const int c0 = tn * BN + wc * CW + 32 * h + 8 * lq;
        if (EPI) {
          const int cl = wc * CW + 32 * h + 8 * lq;
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = act_f(fmaf(v[i], sBN[cl + i], sBN[BN + cl + i]), p.act);
        }
        if (mvalid && c0 < p.Cout && !(DBG && (p.dbg & 8))) {
          if (p.res) add_h8(v, u32x4_t{rv[h].x, rv[h].y, rv[h].z, rv[h].w});
          char* yp = p.y + yoff + c0 * 2;
          if (p.accumulate) add_h8(v, u32x4_t{av[h].x, av[h].y, av[h].z, av[h].w});
          const u32x4_t o = pack_h8(v);
          stg16(yp, uint4{o.x, o.y, o.z, o.w});
          if (BNS) {
            if (sgi >= 0 && c0 < p.bnb.seg[sgi].c1) {
              float gq[8], yf[8];
              Vec<half_t>::unpack(uint4{o.x, o.y, o.z, o.w}, gq);     // gout as stored (what the apply pass reads back)
              Vec<half_t>::unpack(yv[h], yf);
              const int cl = wc * CW + 32 * h + 8 * lq;
              const int act = p.bnb.seg[sgi].act;
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float dz = gq[i] * act_grad_f(fmaf(yf[i], sBN[2 * BN + cl + i], sBN[3 * BN + cl + i]), act);
                st_s[h][i] += dz;
                st_q[h][i] += dz * (yf[i] - sBN[cl + i]) * sBN[BN + cl + i];
              }
            }
          }
        }
      }
    }
    }   // !FUS
  }
  if (FUS) {
    // a workgroup without a tile (the grid is a multiple of eight) still takes part in the barrier; nothing else is left to do
    if (!fus_done) grid_barrier_xcd(p.f_bar, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    return;
  }

  if (BNS ? sgi >= 0 : p.stats != nullptr) {
    // the 16 pixel lanes of a lane group -> one value (DPP row sums), pixel waves -> LDS, one atomic per channel and workgroup
    float* red = reinterpret_cast<float*>(smem);       // [WP][2][BN]  (every tile ended with a barrier; a workgroup without tiles reads nothing)
#pragma unroll
    for (int h = 0; h < CF / 2; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float s = row_sum16(st_s[h][i]), q = row_sum16(st_q[h][i]);
        if (l15 == 15) {
          const int cl = wc * CW + 32 * h + 8 * lq + i;
          red[(wp * 2) * BN + cl] = s; red[(wp * 2 + 1) * BN + cl] = q;
        }
      }
    __syncthreads();
    for (int t = tid; t < 2 * BN; t += NT) {
      const int which = t / BN, cl = t - which * BN;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < WP; ++k) a += red[(k * 2 + which) * BN + cl];
      if (BNS) {
        const BnbSeg& sg = p.bnb.seg[sgi];
        const int Cs = sg.c1 - sg.c0, ci = tn * BN + cl - sg.c0;
        if (ci < Cs) atomicAdd(sg.dsum + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * Cs + which * Cs + ci, a);
      } else {
        const int c = tn * BN + cl;
        if (c < p.Cout) atomicAdd(p.stats + (blockIdx.x % MYOLO_STAT_COPIES) * 2 * p.Cout + which * p.Cout + c, a);
      }
    }
  }
}

template <int BM, int BN, int WP, int WC, int NST, bool DBG = false, bool BNS = false, bool EPI = false, bool BNA = false, bool FUS = false>
int launch(const MidK& k, int per_cu, int ntile_c, hipStream_t st) {
  constexpr int NT = 64 * WP * WC;
  constexpr int SMEM = NST * (BM * (BNA ? 2 : 1) + BN) * 128 + ((BNS || FUS) ? 4 * BN * 4 : (EPI ? 2 * BN * 4 : 0)) + (BNA ? 4 * 512 * 4 : 0);
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(SMEM >= WP * 2 * BN * 4, "statistics reduction area");
  int per_xcd = (256 * per_cu / ntile_c + 7) / 8;
  if (per_xcd < 1) per_xcd = 1;
  if (per_xcd > k.tiles_per_xcd) per_xcd = k.tiles_per_xcd;
  if (FUS) {                        // one tile per workgroup, every workgroup resident (host: tiles_per_xcd * 8 * ntile_c <= 256)
    per_xcd = k.tiles_per_xcd;
    if (per_xcd * 8 * ntile_c > 256) return MYOLO_EINVAL;
  }
  auto kern = conv_mid_kernel<BM, BN, WP, WC, NST, DBG, BNS, EPI, BNA, FUS>;
  MYOLO_ENSURE_DYN_SMEM(kern, SMEM);
  hipLaunchKernelGGL(kern, dim3(per_xcd * 8, ntile_c), dim3(NT), SMEM, st, k);
  MYOLO_CHECK_LAUNCH();
  return 0;
}

}  // namespace mid

static int g_mid_mode = -1;        // 0 off, 1 the layers conv_igemm would run, 2 every qualifying layer (ahead of halo / stream)
static int g_mid_var = 0;          // 0 auto, 1: 128x128 (8 waves), 2: 128x64 (4 waves), 3: 64x128 (4 waves)
static int g_mid_dbg = 0;
static int g_mid_bna_strict = 0;   // tests: myolo_conv_dgrad_bn fails instead of running the two-launch form
static int g_mid_min_tiles = 192;  // 128x128 tiles below this: 64-pixel tiles (twice the workgroups)
static int g_mid_bwd_var = 0;      // experiments: tile variant of the launches WITHOUT forward statistics / eval epilogue (= the dgrads of a training step)
int myolo_conv_mid_set(const char* name, int value) {
  if (!strcmp(name, "mid_mode")) { g_mid_mode = value; return 0; }
  if (!strcmp(name, "mid_var")) { g_mid_var = value; return 0; }
  if (!strcmp(name, "mid_dbg")) { g_mid_dbg = value; return 0; }
  if (!strcmp(name, "mid_bna_strict")) { g_mid_bna_strict = value; return 0; }
  if (!strcmp(name, "mid_min_tiles")) { g_mid_min_tiles = value; return 0; }
  if (!strcmp(name, "mid_bwd_var")) { g_mid_bwd_var = value; return 0; }
  return MYOLO_EINVAL;
}
int myolo_conv_mid_mode() {
  if (g_mid_mode < 0) g_mid_mode = getenv("MYOLO_CONV_MID") ? atoi(getenv("MYOLO_CONV_MID")) : 2;
  return g_mid_mode;
}

// -1: the layer does not qualify (caller goes on to the next kernel family), else 0 / hipError_t.  *bnb_done = 1: the BatchNorm-backward
// sums (myolo_conv_desc.bnb) were produced in the epilogue; 0: the caller runs the reduce pass itself.
// ff != nullptr: the fused forward (myolo_conv_bn_act); dry: only answer whether the fused kernel would run (0) or not (-1)
static int mid_launch(const myolo_conv_desc* d, void* stream, int* bnb_done, const myolo_bn_apply_fold* bf, const myolo_bn_fwd_fuse* ff = nullptr,
                      bool dry = false);
int myolo_conv_mid_try(const myolo_conv_desc* d, void* stream, int* bnb_done) { return mid_launch(d, stream, bnb_done, nullptr); }

static int mid_launch(const myolo_conv_desc* d, void* stream, int* bnb_done, const myolo_bn_apply_fold* bf, const myolo_bn_fwd_fuse* ff, bool dry) {
  using namespace mid;
  *bnb_done = 0;
  if (d->x.dtype != MYOLO_F16 || d->det_no > 0 || d->up_shift != 0) return -1;
  const bool epi = d->scale || d->shift || d->act != MYOLO_ACT_NONE;
  if (epi && (d->stats || (d->bnb && d->nbnb > 0) || bf)) return -1;          // (statistics are taken from the raw accumulators only)
  if (bf) {        // BatchNorm apply in the operand path: 1x1 stride-1 dgrad over dense same-shaped gout / y / dy, <= 512 channels
    auto same = [](const myolo_tensor& a, const myolo_tensor& b) { return a.n == b.n && a.h == b.h && a.w == b.w && a.c == b.c && a.dtype == b.dtype; };
    // (every N tile repeats the transform of its pixel rows: beyond 256 layer channels the two-launch form is faster -- K 512 -> N 512
    //  @16x32x16: 22.5 vs 21.4 us, -> N 1024: 38.1 vs 30.9; profiles/r4_dgrad_bn_ubench.txt)
    constexpr int maxk = 256;        // (round 5 sweep: 128 / 512 measured -0.3 / +0.4 %)
    if (d->ntaps != 1 || d->stride != 1 || d->tap_dy[0] != 0 || d->tap_dx[0] != 0 || d->stats || d->cin_pad > 512 || d->cin_pad > maxk ||
        d->x.c != d->cin_pad) return -1;
    if (!bf->y.ptr || !bf->dy.ptr || !same(d->x, bf->y) || !same(d->x, bf->dy) || d->x.h != d->y.h || d->x.w != d->y.w) return -1;
    if (!bf->saved || !bf->gamma || !bf->beta || !bf->dsum || bf->act != MYOLO_ACT_SILU) return -1;
    if ((bf->y.sw % 8) || (bf->dy.sw % 8) || ((uintptr_t)bf->y.ptr & 15) || ((uintptr_t)bf->dy.ptr & 15)) return -1;
  }
  // (round 6: x.c < cin_pad -- the 48 / 96-channel layers of yolov5m with their weights zero-padded to 64 / 128 input channels -- runs with a
  //  ragged LAST K chunk; MYOLO_MID_RAGGED=0 leaves those layers to conv_igemm / conv_stream as before)
  static const int ragged = getenv("MYOLO_MID_RAGGED") ? atoi(getenv("MYOLO_MID_RAGGED")) : 1;
  if (d->cin_pad % 64 || d->x.c > d->cin_pad || d->cin_pad - d->x.c >= 64 || d->x.c % 8 || (d->x.c != d->cin_pad && !ragged) ||
      d->cout_pad % 64 || d->y.c % 8 || d->ntaps > 25) return -1;
  if (d->res.ptr && (d->res.c < d->y.c)) return -1;
  const int64_t M = (int64_t)d->y.n * d->y.h * d->y.w;
  if (M < 1024 || M > (1 << 24)) return -1;
  // one-step K loops (1x1, 64 channels) are pure streaming: on the 128x256 maps the streaming kernel wins (39.3 vs 43.6 us at batch 16), on
  // the 64x128 maps this one (18.2 -> 13.7)
  if (d->ntaps * (d->cin_pad / 64) < 2 && M > 200000) return -1;      // (200 000: batch 8 of the 128x256 maps too; batch 16 has nothing between 131 072 and 524 288)
  auto extent = [](const myolo_tensor& t) { return ((int64_t)t.n * t.sn + (int64_t)t.h * t.sh + (int64_t)t.w * t.sw + t.c) * 2; };
  if (extent(d->x) >= (1ll << 31) || extent(d->y) >= (1ll << 31) || (d->res.ptr && extent(d->res) >= (1ll << 31))) return -1;
  if ((int64_t)d->cout_pad * d->wtaps * d->cin_pad * 2 >= (1ll << 31)) return -1;
  MidK k;
  k.x = (const char*)d->x.ptr; k.w = (const char*)d->w; k.y = (char*)d->y.ptr; k.res = (const char*)d->res.ptr; k.stats = d->stats;
  k.scale = d->scale; k.shift = d->shift; k.act = d->act;
  k.x_sn = (int)d->x.sn * 2; k.x_sh = (int)d->x.sh * 2; k.x_sw = (int)d->x.sw * 2;
  k.y_sn = (int)d->y.sn * 2; k.y_sh = (int)d->y.sh * 2; k.y_sw = (int)d->y.sw * 2;
  k.r_sn = (int)d->res.sn * 2; k.r_sh = (int)d->res.sh * 2; k.r_sw = (int)d->res.sw * 2;
  k.Hi = d->x.h; k.Wi = d->x.w; k.Wo = d->y.w; k.HWo = d->y.h * d->y.w; k.M = (int)M; k.Cout = d->y.c;
  k.stride = d->stride; k.ntaps = d->ntaps; k.kchunks = d->cin_pad / 64; k.nsteps = k.ntaps * k.kchunks;
  k.wrow_bytes = d->wtaps * d->cin_pad * 2;
  k.cin_b = d->x.c * 2;
  k.accumulate = d->accumulate; k.dbg = g_mid_dbg;
  k.by = nullptr; k.bdy = nullptr; k.b_K = 0;
  if (bf) {
    if (extent(bf->y) >= (1ll << 31) || extent(bf->dy) >= (1ll << 31)) return -1;
    k.by = (const char*)bf->y.ptr; k.bdy = (char*)bf->dy.ptr;
    k.by_sn = (int)bf->y.sn * 2; k.by_sh = (int)bf->y.sh * 2; k.by_sw = (int)bf->y.sw * 2;
    k.dy_sn = (int)bf->dy.sn * 2; k.dy_sh = (int)bf->dy.sh * 2; k.dy_sw = (int)bf->dy.sw * 2;
    k.b_saved = bf->saved; k.b_gamma = bf->gamma; k.b_beta = bf->beta; k.b_dsum = bf->dsum; k.b_dgamma = bf->dgamma; k.b_dbeta = bf->dbeta;
    k.b_act = bf->act; k.b_K = d->x.c; k.b_rM = 1.0f / (float)M;
    k.dbg = 0;
  }
  for (int t = 0; t < MYOLO_MAX_TAPS; ++t) {
    const bool in = t < d->ntaps;
    k.tap_dy[t] = in ? d->tap_dy[t] : 0; k.tap_dx[t] = in ? d->tap_dx[t] : 0;
    k.tap_xoff[t] = in ? d->tap_dy[t] * k.x_sh + d->tap_dx[t] * k.x_sw : 0;
    k.tap_woff[t] = in ? d->tap_w[t] * d->cin_pad * 2 : 0;
  }
  const int bn = d->cout_pad % 128 == 0 ? 128 : 64;
  const int ntile_c = d->cout_pad / bn;
  int var = g_mid_var;
  if (!var && g_mid_bwd_var && !d->stats && !epi && bn == 128) var = g_mid_bwd_var;
  // round 5, yolov5m's widths (models/yolov5m_city_seg.yaml: 192 / 384 channels): a 192-wide N tile instead of three 64-wide ones that each
  // stage the same pixel rows again (37 + 23 launches of the yolov5m + Lab step ran <128, 64> tiles); only with at least one tile per CU
  const bool wide192 = !bf && (d->cout_pad == 192 || d->cout_pad == 384 || d->cout_pad == 576) &&
                       ((M + 127) / 128) * (d->cout_pad / 192) >= 256;
  if (var == 0 && wide192) var = 6;
  if (var == 6 && (d->cout_pad % 192 || bf)) var = 0;
  if (var == 0) {
    var = bn == 64 ? 2 : (((M + 127) / 128) * ntile_c < g_mid_min_tiles ? 3 : 1);
    // K-heavy layers with >= 2 tiles of 256 x 128 per CU: 64 x 64 wave tiles (2/3 of the LDS and L2 bytes per MFMA; the segmentation
    // head's 3x3 256 -> 128 at 64x128: 106 -> 94 us, its dgrad 114 -> 103; scripts/conv_train_ubench.py PROBE=mid)
    if (var == 1 && ((M + 255) / 256) * ntile_c >= 512 && d->ntaps * (d->cin_pad / 64) >= 16) var = 4;
  }
  if (bn == 64 && var != 6) var = 2;
  if (ff || dry) {
    // fused forward: every tile of the layer resident at ONE workgroup per CU with its accumulators held -> at most 256 tiles (padded to
    // a multiple of eight per N tile).  The variant the plain launch would take if that fits, else the 256-row tile, else not fused
    auto wgs = [&](int v) {
      const int bm_ = v == 3 ? 64 : (v == 4 ? 256 : 128);
      const int bn_ = v == 2 ? 64 : 128;
      return (int64_t)(((M + bm_ - 1) / bm_ + 7) / 8) * 8 * (d->cout_pad / bn_);
    };
    if (var == 6) return -1;
    if (wgs(var) > 256) {
      if (bn == 128 && wgs(4) <= 256) var = 4;
      else return -1;
    }
    if (dry) return 0;
  }
  const int bm = var == 3 ? 64 : (var == 4 ? 256 : 128);
  k.ntile_p = (int)((M + bm - 1) / bm);
  k.tiles_per_xcd = (k.ntile_p + 7) / 8;
  if (var == 1 && k.nsteps < 3) var = 5;               // (short K loops: the three-stage ring)
  if (bf) var = (var == 2 || var == 3) ? var : 5;      // (two pixel tiles per stage: three stages fill the LDS)
  hipStream_t st = (hipStream_t)stream;
  const int bn_eff = var == 2 ? 64 : (var == 6 ? 192 : 128), ntc = d->cout_pad / bn_eff;
  // the BatchNorm-backward sums of the layer(s) below ride in the epilogue when every segment is a whole number of N tiles
  const bool fold = d->bnb && d->nbnb > 0 && d->nbnb <= MYOLO_MAX_BNB && !d->stats && !k.dbg && bnb_aligned(d, bn_eff);
  k.bnb.n = 0;
  if (fold) { bnb_fill(&k.bnb, d); *bnb_done = 1; }
  if (bf) {
    if (var == 5) return fold ? launch<128, 128, 4, 2, 3, false, true, false, true>(k, 1, ntc, st) : launch<128, 128, 4, 2, 3, false, false, false, true>(k, 1, ntc, st);
    if (var == 2) return fold ? launch<128, 64, 2, 2, 3, false, true, false, true>(k, 1, ntc, st) : launch<128, 64, 2, 2, 3, false, false, false, true>(k, 1, ntc, st);
    return fold ? launch<64, 128, 1, 4, 3, false, true, false, true>(k, 1, ntc, st) : launch<64, 128, 1, 4, 3, false, false, false, true>(k, 1, ntc, st);
  }
  if (ff) {
    const myolo_bn_split* sp = ff->split;
    k.f_out = (char*)ff->out.ptr; k.f_res = (const char*)ff->res.ptr;
    k.fo_sn = (int)ff->out.sn * 2; k.fo_sh = (int)ff->out.sh * 2; k.fo_sw = (int)ff->out.sw * 2;
    k.fr_sn = (int)ff->res.sn * 2; k.fr_sh = (int)ff->res.sh * 2; k.fr_sw = (int)ff->res.sw * 2;
    k.f_gamma = ff->gamma; k.f_beta = ff->beta; k.f_rm = ff->running_mean; k.f_rv = ff->running_var; k.f_nbt = ff->nbt; k.f_saved = ff->saved;
    k.f_eps = ff->eps; k.f_mom = ff->momentum; k.f_act = ff->act; k.f_bar = ff->barrier;
    k.f_cs = sp ? sp->c_split : d->y.c;
    k.f_world = (sp && sp->count_scale > 1) ? sp->count_scale : 1;
    const bool two = sp && sp->c_split < d->y.c;
    k.f_gamma2 = two ? sp->gamma2 : ff->gamma; k.f_beta2 = two ? sp->beta2 : ff->beta;
    k.f_rm2 = two ? sp->running_mean2 : nullptr; k.f_rv2 = two ? sp->running_var2 : nullptr; k.f_nbt2 = two ? sp->nbt2 : nullptr;
    if (var == 1) return launch<128, 128, 4, 2, 4, false, false, false, false, true>(k, 1, ntc, st);
    if (var == 4) return launch<256, 128, 4, 2, 3, false, false, false, false, true>(k, 1, ntc, st);
    if (var == 5) return launch<128, 128, 4, 2, 3, false, false, false, false, true>(k, 1, ntc, st);
    if (var == 2) return launch<128, 64, 2, 2, 3, false, false, false, false, true>(k, 1, ntc, st);
    return launch<64, 128, 1, 4, 3, false, false, false, false, true>(k, 1, ntc, st);
  }
  if (var == 1 && k.dbg) return launch<128, 128, 4, 2, 4, true>(k, 1, ntc, st);     // profiling switches (myolo_set_option("mid_dbg", bits))
#define MID_GO(BM_, BN_, WP_, WC_, NST_, PCU_)                                                   \
  return fold ? launch<BM_, BN_, WP_, WC_, NST_, false, true>(k, PCU_, ntc, st)                   \
              : (epi ? launch<BM_, BN_, WP_, WC_, NST_, false, false, true>(k, PCU_, ntc, st) : launch<BM_, BN_, WP_, WC_, NST_>(k, PCU_, ntc, st))
  if (var == 1) MID_GO(128, 128, 4, 2, 4, 1);       // four stages: loads three K steps ahead
  if (var == 4) MID_GO(256, 128, 4, 2, 3, 1);       // 64 x 64 wave tiles
  if (var == 5) MID_GO(128, 128, 4, 2, 3, 1);       // three stages
  if (var == 2) MID_GO(128, 64, 2, 2, 3, 2);
  if (var == 6) MID_GO(128, 192, 4, 2, 3, 1);       // 192-wide N tile (yolov5m)
  MID_GO(64, 128, 1, 4, 3, 2);
#undef MID_GO
}

// ---- 1x1 dgrad with the BatchNorm-backward apply pass in its operand path (myolo.h) ----
extern "C" int myolo_conv_dgrad_bn(const myolo_conv_desc* d, const myolo_bn_apply_fold* f, void* stream) {
  if (!d || !f || !d->x.ptr || !d->y.ptr || !d->w || !f->y.ptr || !f->dy.ptr) return MYOLO_EINVAL;
  static const int off = getenv("MYOLO_BN_APPLY_FOLD") ? atoi(getenv("MYOLO_BN_APPLY_FOLD")) == 0 : 0;
  const bool want_bnb = d->bnb != nullptr && d->nbnb > 0;
  if (!off && d->x.dtype == MYOLO_F16 && myolo_conv_mid_mode() > 0) {
    int done = 0;
    const int r = mid_launch(d, stream, &done, f);
    if (r != -1) {
      if (r) return r;
      return (want_bnb && !done) ? myolo_bnb_fallback(d, &d->y, stream) : 0;
    }
  }
  // not foldable (fp32, more than 512 channels, strided views ...): the apply pass as its own launch, then the plain dgrad over dy
  if (g_mid_bna_strict) return MYOLO_EINVAL;
  myolo_tensor none{};
  int r = myolo_bn_act_bwd_apply(&d->x, &f->y, f->saved, f->gamma, f->beta, f->act, f->dsum, f->dgamma, f->dbeta, &f->dy, &none, 0, stream);
  if (r) return r;
  myolo_conv_desc d2 = *d;
  d2.x = f->dy;
  return myolo_conv(&d2, stream);
}

// ---- training-mode Conv + BatchNorm + activation in ONE launch (myolo.h, round 6) ----
static int g_conv_bn_act = -1;         // MYOLO_CONV_BN_ACT=0: always the two launches
static bool conv_bn_act_pre(const myolo_conv_desc* d) {
  if (g_conv_bn_act < 0) g_conv_bn_act = getenv("MYOLO_CONV_BN_ACT") ? atoi(getenv("MYOLO_CONV_BN_ACT")) : 1;
  return g_conv_bn_act && d && d->x.ptr && d->y.ptr && d->w && d->x.dtype == MYOLO_F16 && d->y.dtype == MYOLO_F16 && d->stats && !d->res.ptr &&
         !d->accumulate && !d->scale && !d->shift && d->act == MYOLO_ACT_NONE && !(d->bnb && d->nbnb > 0) && d->det_no == 0 &&
         myolo_conv_mid_mode() >= 2 && !g_mid_var && !g_mid_dbg &&
         !(d->ntaps > 1 && d->stride == 1);      // (k x k stride-1 layers: conv_midx's resident-input kernel runs them as the first of two launches)
}
extern "C" int myolo_conv_bn_act_ok(const myolo_conv_desc* d) {
  if (!conv_bn_act_pre(d)) return 0;
  int done = 0;
  return mid_launch(d, nullptr, &done, nullptr, nullptr, true) == 0;
}
extern "C" int myolo_conv_bn_act(const myolo_conv_desc* d, const myolo_bn_fwd_fuse* f, void* stream) {
  if (!d || !f || !d->stats || !f->gamma || !f->beta || !f->out.ptr || !f->barrier || ((uintptr_t)f->barrier & 127)) return MYOLO_EINVAL;
  auto same = [](const myolo_tensor& a, const myolo_tensor& b) { return a.n == b.n && a.h == b.h && a.w == b.w && a.c == b.c && a.dtype == b.dtype; };
  auto extent = [](const myolo_tensor& t) { return ((int64_t)t.n * t.sn + (int64_t)t.h * t.sh + (int64_t)t.w * t.sw + t.c) * 2; };
  if (!same(f->out, d->y) || (f->res.ptr && (f->res.n != d->y.n || f->res.h != d->y.h || f->res.w != d->y.w || f->res.c < d->y.c || f->res.dtype != d->y.dtype)))
    return MYOLO_EINVAL;
  const bool vec = !(f->out.sw % 8) && !(f->out.sh % 8) && !(f->out.sn % 8) && !((uintptr_t)f->out.ptr & 15) && extent(f->out) < (1ll << 31) &&
                   (!f->res.ptr || (!(f->res.sw % 8) && !(f->res.sh % 8) && !(f->res.sn % 8) && !((uintptr_t)f->res.ptr & 15) && extent(f->res) < (1ll << 31)));
  const bool split_ok_ = !f->split || (f->split->c_split > 0 && f->split->c_split <= d->y.c && !(f->split->c_split % 8) &&
                                       (f->split->c_split == d->y.c || (f->split->gamma2 && f->split->beta2)));
  if (!split_ok_) return MYOLO_EINVAL;
  if (vec && conv_bn_act_pre(d)) {
    int done = 0;
    const int r = mid_launch(d, stream, &done, nullptr, f, false);
    if (r != -1) return r;
  }
  // the definition of the result: the two launches
  int r = myolo_conv(d, stream);
  if (r) return r;
  return myolo_bn_act_fwd_split(&d->y, d->stats, f->gamma, f->beta, f->running_mean, f->running_var, f->nbt, f->saved, f->eps, f->momentum, f->act,
                                f->res.ptr ? &f->res : nullptr, &f->out, f->split, stream);
}

// mkws_embed_rows.hip -- whole-MBConv kernel for the big-image blocks in which the depthwise output never leaves the REGISTERS of
// the waves that produce it (round 6; blocks 2b and 3b of the network defined at multilingual_kws/train_multilingual_embedding.py:58-83).
//
// mbconv_mid_kernel keeps a clip's depthwise output D [HoWo, Cexp] in LDS (2b: 77 KB of its 132 KB): ONE workgroup per CU, whose
// barrier-separated phases (expand / depthwise / SE / project) have nothing to overlap with -- 17.1 us per clip against an issue floor of
// ~7 us (profiles/r04_notes.md section 2).  Here a WAVE owns one 16-row tile of a clip for the whole block:
//
//   per 16-channel chunk j of the expanded tensor (Cexp / 16 chunks, fully unrolled):
//     expand    the wave's input row tile(s) x chunk j on the MFMA (weights = A operand fragments staged in LDS, block input = B operand
//               fragments held in registers for the whole block) + BN + swish -> E[j & 1] in LDS: the clip's image with a ZERO BORDER, one
//               16-channel slab per position, so that every depthwise tap is a ds_read_b128 at base + immediate (no bounds arithmetic)
//     barrier   (ONE per chunk: E, the expand constants and the depthwise constants are double-buffered)
//     depthwise lane (g, c) = output position 16 m + c, channels 16 j + 4 g .. + 3: KS x KS taps from E and the staged tap table, BN, swish
//               -> D[j], a float4 that IS this lane's B-operand fragment of the projection (k = 16 j + 4 g + s in step s)
//   SE          column sums by DPP row reductions (the 16 positions of a tile are the 16 lanes of a DPP row) + one LDS partial per tile,
//               tiles of a clip added in tile order; the two FCs as in mbconv_mid_kernel (thread = (unit, channel slice), fixed-order folds)
//   project     D[j] * gate (registers x LDS float4) against the projection weights streamed global / L2 -> registers; BN, residual, store
//
// LDS per workgroup: 43 KB (2b) / 57 KB (3b, two clips) instead of 132 KB, so 2-3 workgroups share a CU and one clip's barriers and SE
// latency hide under another's MFMA work.  Row tiles never cross clips (a clip's SE sums have one order whatever its slot in the
// workgroup): results are bit-identical across batch sizes and compositions, like every other kernel of the library.
// Constants are staged global -> register -> LDS by role threads, the expand set two chunks ahead and the depthwise set one chunk ahead
// (windows between barriers in which nobody reads the target buffer: see the schedule in the kernel).
#include "mkws_common.h"

#include <cstdlib>
#include <type_traits>
#include <vector>

#pragma clang fp contract(fast)
#include "mkws_embed_dev.h"
#include "mkws_embed_rows.h"

#ifdef MKWS_FRONT_TIMING
// timing build: every wave stamps s_memtime at the phase boundaries (scalar registers; wave 0 reports); MKWS_ABLATE bits change / skip parts (1: tap-table reads at per-lane addresses instead of broadcast ones,
// 4: expand MFMAs + epilogue, 8: staging, 16: SE phase, 32: projection)
#define ROWS_T_DECL long long rt_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long rt_t_ = clock64(); const long long rt_t0_ = rt_t_
#define ROWS_T(k) do { const long long n_ = clock64(); rt_acc_[k] += n_ - rt_t_; rt_t_ = n_; } while (0)   /* wave-uniform: stays on the scalar unit */
#define ROWS_T_STORE() do { if (threadIdx.x == 0 && a.dbg_t) { unsigned long long* p_ = a.dbg_t + (size_t)blockIdx.x * 8; \
    for (int k_ = 0; k_ < 7; ++k_) p_[k_] = (unsigned long long)rt_acc_[k_]; p_[7] = (unsigned long long)(clock64() - rt_t0_); } } while (0)
#define ROWS_ABL(bit) (a.ablate & (bit))
#else
#define ROWS_T_DECL
#define ROWS_T(k)
#define ROWS_T_STORE()
#define ROWS_ABL(bit) false
#endif

namespace mkws {

namespace {

constexpr int cmax_(int a, int b) { return a > b ? a : b; }

template <int KS, int S, int KCT, int HT, int WT, int CEXP, int NTP, int G, int CPW, int LDE, int SWZ, int PARK>
struct RowsGeom {
  static constexpr int HW = HT * WT;
  static constexpr int HoT = (S == 1) ? HT : (HT + 1) / 2, WoT = (S == 1) ? WT : (WT + 1) / 2;
  static constexpr int HoWo = HoT * WoT;
  // Keras padding: "same" for stride 1, correct_pad + "valid" for stride 2
  static constexpr int PT = (S == 1) ? KS / 2 : KS / 2 - (1 - HT % 2), PLF = (S == 1) ? KS / 2 : KS / 2 - (1 - WT % 2);
  static constexpr int HP = PT + cmax_(HT, (HoT - 1) * S - PT + KS), WP = PLF + cmax_(WT, (WoT - 1) * S - PLF + KS);   // image + zero border
  static constexpr int NPC = HP * WP, NPOS = G * NPC;
  static constexpr int MTC = (HoWo + 15) / 16;                    // output row tiles of a clip = waves of a clip group
  static constexpr int MTIC = (HW + 15) / 16;                     // input row tiles of a clip
  static constexpr int TI = (MTIC + MTC - 1) / MTC;               // input row tiles (per clip) a wave expands
  static constexpr int GW = G / CPW;                              // clip groups: a wave owns row tile m of the CPW clips of its group
  static constexpr int NW = GW * MTC, NTHR = 64 * NW;
  static constexpr int KC = CEXP / 16;
  static constexpr int NEXP = KCT * 64 + 8, NDW = KS * KS * 4 + 8; // staged float4s per chunk: [expand fragments | scE | shE], [taps | scD | shD]
  static constexpr int NSL = NTHR / 16, CPS = (CEXP + NSL - 1) / NSL;
  static constexpr int ESZ = NPOS * LDE;                          // floats of one E buffer
  static constexpr int oE = 0;
  static constexpr int oSexp = oE + 2 * ESZ;
  static constexpr int oSdw = oSexp + 2 * NEXP * 4;
  static constexpr int oPart = oSdw + 2 * NDW * 4;               // [G][MTC][CEXP] column sums of each (clip, tile)
  static constexpr int oMean = oPart + G * MTC * CEXP;           // [G][CEXP]
  static constexpr int oGate = oMean + G * CEXP;                 // [G][CEXP]
  static constexpr int oFc1 = oGate + G * CEXP;                  // [NSL][16][G]
  static constexpr int oR = oFc1 + NSL * 16 * G;                 // [G][16]
  static constexpr int oPark = oR + G * 16;                      // [NW][CPW][PARK][64] float4: the LAST `PARK` chunks of D, lane-private (written and read back by the same lane: no barrier)
  static constexpr int lds_floats = oPark + NW * CPW * PARK * 256;
  static_assert(CEXP % 16 == 0 && LDE % 4 == 0 && LDE >= 16 && PARK >= 0 && PARK < CEXP / 16, "slabs are whole MFMA tiles");
  static_assert(G % CPW == 0 && (CPW == 1 || !SWZ || NPC % 4 == 0), "clips of a wave sit a whole number of slab rotations apart");
  static_assert(NEXP + NDW <= NTHR, "one staging role per thread");
  static_assert(NTHR >= CEXP && NTHR >= 16 * G && NTHR >= G * CEXP / 4 && NTHR <= 1024, "thread roles of the SE phase");
  static_assert((size_t)lds_floats * 4 <= 160 * 1024, "LDS carve exceeds one CU");
};

// sum over the 16 lanes of a DPP row, left in every lane of the row (each step adds a permutation of the row to itself: commutative at
// every level, so all 16 lanes hold the same bits)
__device__ __forceinline__ float row16_sum_(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
  return v;
}

template <int KS, int S, int KCT, int HT, int WT, int CEXP, int NTP, int G, int CPW, int LDE, int SWZ, int PARK, int WPE>
__global__ __launch_bounds__((RowsGeom<KS, S, KCT, HT, WT, CEXP, NTP, G, CPW, LDE, SWZ, PARK>::NTHR), WPE) void mbconv_rows_kernel(MidArgs a) {
  using GM = RowsGeom<KS, S, KCT, HT, WT, CEXP, NTP, G, CPW, LDE, SWZ, PARK>;
  constexpr int HW = GM::HW, WoT = GM::WoT, HoWo = GM::HoWo, PT = GM::PT, PLF = GM::PLF, WP = GM::WP, NPC = GM::NPC;
  constexpr int MTC = GM::MTC, MTIC = GM::MTIC, TI = GM::TI, NTHR = GM::NTHR, KC = GM::KC, NEXP = GM::NEXP, NDW = GM::NDW;
  constexpr int NSL = GM::NSL, CPS = GM::CPS, ESZ = GM::ESZ;
  constexpr int SE_MAX = 10;                                      // SE units of blocks 2a..4a: 4, 6, 6, 10, 10
  constexpr int NA = SWZ ? 4 : 1;                                 // distinct slab rotations a lane's taps meet
  constexpr int PD = 4;                                           // depth of the projection weight ring (chunks): L2 hits, every wave of every workgroup streams the same fragments
  constexpr int KR = KC - PARK;                                   // chunks of D that stay in registers
  constexpr int CLIP = NPC * LDE;                                 // floats between the images of consecutive clips in E
  extern __shared__ __attribute__((aligned(16))) float s_rows[];
  float* s_E = s_rows + GM::oE;
  float* s_Sexp = s_rows + GM::oSexp;
  float* s_Sdw = s_rows + GM::oSdw;
  float* s_part = s_rows + GM::oPart;
  float* s_mean = s_rows + GM::oMean;
  float* s_gate = s_rows + GM::oGate;
  float* s_fc1 = s_rows + GM::oFc1;
  float* s_r = s_rows + GM::oR;
  float* s_park = s_rows + GM::oPark;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // uniform by construction
  const int g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * G;
  const int gvalid = (a.B - b0 < G) ? (a.B - b0) : G;
  const int gw = wave / MTC, m = wave - gw * MTC;                 // this wave's clip group and row tile; its clips: slots gi0 .. gi0 + CPW - 1
  const int gi0 = gw * CPW;

  // ---- staging roles: thread e < NEXP moves float4 e of the expand set, NEXP <= e < NEXP + NDW float4 e - NEXP of the depthwise set ----
  const bool role_exp = tid < NEXP, role_dw = tid >= NEXP && tid < NEXP + NDW;
  const int lead = role_exp ? 2 : 1;                              // chunks ahead of the loop at which this role STORES
  const float* st_src = a.scD; int st_stride = 16;                // source of chunk 0, floats per chunk
  float* st_dst; int st_flip;                                     // destination in buffer 0, floats between the two buffers
  {
    int e = tid;
    if (e < KCT * 64) { st_src = a.WpE + ((size_t)(e >> 6) * a.NTtotE * 64 + (e & 63)) * 4; st_stride = 256; }
    else if (e < NEXP) { const int q = e - KCT * 64; st_src = (q < 4 ? a.scE + 4 * q : a.shE + 4 * (q - 4)); st_stride = 16; }
    else {
      e -= NEXP;
      if (e < KS * KS * 4) { st_src = a.Wd + (size_t)(e >> 2) * CEXP + 4 * (e & 3); st_stride = 16; }
      else { const int q = (e - KS * KS * 4) & 7; st_src = (q < 4 ? a.scD + 4 * q : a.shD + 4 * (q - 4)); st_stride = 16; }
    }
    st_dst = role_exp ? s_Sexp + 4 * tid : s_Sdw + 4 * (role_dw ? tid - NEXP : 0);
    st_flip = role_exp ? NEXP * 4 : NDW * 4;
  }
  auto st_load = [&](int j) { return *reinterpret_cast<const f32x4*>(st_src + (size_t)j * st_stride); };
  auto st_store = [&](int j, const f32x4& v) { *reinterpret_cast<f32x4*>(st_dst + (j & 1) * st_flip) = v; };
  f32x4 st = {0.f, 0.f, 0.f, 0.f};

  // ---- prologue: zero both E buffers (the border must read 0), expand sets of chunks 0 and 1, depthwise set of chunk 0 ----
  if (role_exp || role_dw) st = st_load(0);
  for (int i = tid; i < 2 * ESZ / 4; i += NTHR) reinterpret_cast<f32x4*>(s_E)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (role_exp || role_dw) st_store(0, st);
  if (role_exp && KC > 1) { st = st_load(1); st_store(1, st); }
  if ((role_exp && KC > 2) || (role_dw && KC > 1)) st = st_load(lead);      // in flight across the first barrier

  // block input rows of this wave as B-operand fragments (k = 16 kc + 4 g .. + 3 of row 16 mt + c of each of its clips); where they land in E
  // (the clips of a wave sit CLIP floats apart, a whole number of slab rotations: one address per input tile serves all of them)
  f32x4 x[CPW][TI][KCT];
  int ew[TI];
  bool in_row[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int mt = m + MTC * ti;
    const int qi = 16 * mt + c;
    in_row[ti] = mt < MTIC && qi < HW;
    const int qc = qi < HW ? qi : HW - 1;
    const int ih = qc / WT, iw = qc - ih * WT;
    const int idx = gi0 * NPC + (ih + PT) * WP + iw + PLF;
    ew[ti] = idx * LDE + 4 * ((g + idx * SWZ) & 3);
#pragma unroll
    for (int cw = 0; cw < CPW; ++cw)
#pragma unroll
      for (int kc = 0; kc < KCT; ++kc) {
        x[cw][ti][kc] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (in_row[ti] && gi0 + cw < gvalid && 16 * kc + 4 * g < a.Cin)
          x[cw][ti][kc] = *reinterpret_cast<const f32x4*>(a.X + ((size_t)(b0 + gi0 + cw) * HW + qi) * a.Cin + 16 * kc + 4 * g);
      }
  }
  // output position of this lane and the LDS addresses its taps start from: tap (i, jx) of the padded image sits cst = i * WP + jx slabs
  // behind the top-left one; with the slab rotation (SWZ) its channel quad is quad (g + idx) & 3 of the slab, so four bases cover all taps
  const int q = 16 * m + c;
  const bool out_row = q < HoWo;
  const float* etap[NA];
  {
    const int qc = q < HoWo ? q : HoWo - 1;
    const int oh = qc / WoT, ow = qc - oh * WoT;
    const int pb = gi0 * NPC + oh * S * WP + ow * S;
#pragma unroll
    for (int k = 0; k < NA; ++k) etap[k] = s_E + pb * LDE + 4 * ((g + (pb + k) * SWZ) & 3);
  }
  const float* wexp = s_Sexp + 4 * lane;                          // + (kc * 64) * 4 + buffer
  const float* cexp = s_Sexp + KCT * 256 + 4 * g;                 // scE; shE 16 floats behind
  float* park = s_park + (wave * CPW * (PARK > 0 ? PARK : 1) * 64 + lane) * 4;   // + (cw * PARK + pj) * 256
  const float* wtap = s_Sdw + 4 * g + (ROWS_ABL(1) ? 16 * c : 0);  // + t * 16 + buffer   (timing build, MKWS_ABLATE & 1: every lane its own address instead of 16 lanes per address)
  ROWS_T_DECL;
  __syncthreads();
  ROWS_T(0);

  f32x4 D[CPW][KR > 0 ? KR : 1];
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    const int eb = (j & 1) * ESZ, sb = (j & 1) * NEXP * 4, db = (j & 1) * NDW * 4;
    // ---- expand chunk j of this wave's input rows -> E[j & 1]: CPW * TI independent accumulator chains share the weight fragments ----
    if (!ROWS_ABL(4)) {
      const f32x4 sce = *reinterpret_cast<const f32x4*>(cexp + sb), she = *reinterpret_cast<const f32x4*>(cexp + sb + 16);
      f32x4 w[KCT];
#pragma unroll
      for (int kc = 0; kc < KCT; ++kc) w[kc] = *reinterpret_cast<const f32x4*>(wexp + sb + kc * 256);
      f32x4 acc[CPW][TI];
#pragma unroll
      for (int cw = 0; cw < CPW; ++cw)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) acc[cw][ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KCT; ++kc)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int cw = 0; cw < CPW; ++cw)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) acc[cw][ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kc][s4], x[cw][ti][kc][s4], acc[cw][ti], 0, 0, 0);
#pragma unroll
      for (int cw = 0; cw < CPW; ++cw)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
          const f32x4 y = swish4_(acc[cw][ti] * sce + she);
          if (in_row[ti] && gi0 + cw < gvalid) *reinterpret_cast<f32x4*>(s_E + eb + ew[ti] + cw * CLIP) = y;
        }
    }
    ROWS_T(1);
    __syncthreads();
    ROWS_T(2);
    // ---- staging window (barrier j, barrier j + 1): nobody reads the depthwise set of chunk j + 1's buffer (its last readers were the
    //      depthwise of chunk j - 1) nor the expand set of chunk j + 2's buffer (last read by the expand of chunk j) ----
    if ((role_exp || role_dw) && j + lead < KC && !ROWS_ABL(8)) {
      st_store(j + lead, st);
      if (j + lead + 1 < KC) st = st_load(j + lead + 1);
    }
    // ---- depthwise of chunk j for this lane's output position in each of its clips (the tap table is read once for all of them) ----
    {
      f32x4 acc[CPW];
#pragma unroll
      for (int cw = 0; cw < CPW; ++cw) acc[cw] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        // one tap row at a time: its (CPW + 1) KS ds_reads go out together, the multiply-adds follow in tap order, and the scheduler may not
        // pull the next row's reads up (all KS x KS at once = 4 (CPW + 1) KS^2 live registers: spills at the occupancy this kernel is built for)
        f32x4 v[CPW][KS], wv[KS];
#pragma unroll
        for (int jx = 0; jx < KS; ++jx) {
          const int cst = i * WP + jx;
#pragma unroll
          for (int cw = 0; cw < CPW; ++cw) v[cw][jx] = *reinterpret_cast<const f32x4*>(etap[SWZ ? (cst & 3) : 0] + eb + cst * LDE + cw * CLIP);
          wv[jx] = *reinterpret_cast<const f32x4*>(wtap + db + (i * KS + jx) * 16);
        }
#pragma unroll
        for (int jx = 0; jx < KS; ++jx)
#pragma unroll
          for (int cw = 0; cw < CPW; ++cw) acc[cw] += v[cw][jx] * wv[jx];
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4 scd = *reinterpret_cast<const f32x4*>(wtap + db + KS * KS * 16), shd = *reinterpret_cast<const f32x4*>(wtap + db + KS * KS * 16 + 16);
#pragma unroll
      for (int cw = 0; cw < CPW; ++cw) {
        f32x4 y = swish4_(acc[cw] * scd + shd);
        const bool ok = out_row && gi0 + cw < gvalid;
        if (!ok) y = (f32x4){0.f, 0.f, 0.f, 0.f};                  // padding rows: out of the SE sums; their projection is never stored
        asm volatile("" : "+v"(y));                                // computed HERE: without this hipcc sinks every chunk's multiply-adds behind the loop (to D's first use)
                                                                   // and carries the 2 KS^2 loaded float4s of all chunks there through scratch
        // SE squeeze, this tile's share: the 16 positions of the tile are the 16 lanes of a DPP row
        {
          f32x4 t = y;
          t.x = row16_sum_(t.x); t.y = row16_sum_(t.y); t.z = row16_sum_(t.z); t.w = row16_sum_(t.w);
          if (c == 0) *reinterpret_cast<f32x4*>(s_part + ((gi0 + cw) * MTC + m) * CEXP + 16 * j + 4 * g) = t;
        }
        if (a.dbg_dw && ok) *reinterpret_cast<f32x4*>(a.dbg_dw + ((size_t)(b0 + gi0 + cw) * HoWo + q) * CEXP + 16 * j + 4 * g) = y;
        if (j < KR) D[cw][j < KR ? j : 0] = y;
        else *reinterpret_cast<f32x4*>(park + (cw * PARK + j - KR) * 256) = y;   // the last PARK chunks wait in LDS (lane-private)
      }
    }
    ROWS_T(3);
  }

  // ---- the SE weights of this thread's roles and the head of the projection stream: requested now, used behind the barriers below ----
  float wr_pre[CPS], we_pre[SE_MAX], br_pre, be_pre;
  {
    // (unconditional loads at clamped indices, masked afterwards: a branch per element costs more than the load)
    const int n = tid & 15, sl = tid >> 4;
    const int nc = n < a.se ? n : a.se - 1;
#pragma unroll
    for (int i = 0; i < CPS; ++i) {
      const int ch = sl * CPS + i;
      float w = a.Wr[(size_t)(ch < CEXP ? ch : CEXP - 1) * a.se + nc];
      asm volatile("" : "+v"(w));                                  // (keeps the load where it is: hipcc otherwise sinks each one into its own branch, with its own s_waitcnt vmcnt(0))
      wr_pre[i] = (ch < CEXP && n < a.se) ? w : 0.0f;
    }
    const int tc = tid < CEXP ? tid : CEXP - 1;
#pragma unroll
    for (int n2 = 0; n2 < SE_MAX; ++n2) {
      float w = a.We[(size_t)(n2 < a.se ? n2 : a.se - 1) * CEXP + tc];
      asm volatile("" : "+v"(w));
      we_pre[n2] = (tid < CEXP && n2 < a.se) ? w : 0.0f;
    }
    const int nb = tid / G;
    br_pre = a.br[nb < a.se ? nb : a.se - 1];
    be_pre = a.be[tc];
  }
  const WBuf p_w(a.WpP, (unsigned)(lane * 4));
  f32x4 wq[PD][NTP];
#pragma unroll
  for (int d = 0; d < PD; ++d)
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) wq[d][nt] = p_w.ld((size_t)((d < KC ? d : KC - 1) * NTP + nt) * 256);
  ROWS_T(4);
  if (!ROWS_ABL(16)) {
  // ---- SE squeeze: the tiles' column sums were left in s_part chunk by chunk; the clip's tiles in tile order ----
  __syncthreads();
  if (tid < G * (CEXP / 4)) {
    const int gi2 = tid / (CEXP / 4), q4 = tid - gi2 * (CEXP / 4);
    f32x4 t = *reinterpret_cast<const f32x4*>(s_part + (gi2 * MTC) * CEXP + 4 * q4);
#pragma unroll
    for (int mm = 1; mm < MTC; ++mm) t += *reinterpret_cast<const f32x4*>(s_part + (gi2 * MTC + mm) * CEXP + 4 * q4);
    *reinterpret_cast<f32x4*>(s_mean + gi2 * CEXP + 4 * q4) = t * (1.0f / (float)HoWo);
  }
  __syncthreads();
  // ---- SE reduce: thread (unit n, channel slice sl) folds its CPS channels; slices are then added in fixed order ----
  {
    const int n = tid & 15, sl = tid >> 4;
#pragma unroll
    for (int gi2 = 0; gi2 < G; ++gi2) {
      float v = 0.0f;
#pragma unroll
      for (int i = 0; i < CPS; ++i) {
        const int ch = sl * CPS + i;
        v += s_mean[gi2 * CEXP + (ch < CEXP ? ch : 0)] * wr_pre[i];
      }
      s_fc1[(sl * 16 + n) * G + gi2] = v;
    }
  }
  __syncthreads();
  if (tid < 16 * G) {
    const int n = tid / G, gi2 = tid - n * G;
    float v = 0.0f;
#pragma unroll 4
    for (int sl = 0; sl < NSL; ++sl) v += s_fc1[(sl * 16 + n) * G + gi2];
    s_r[gi2 * 16 + n] = (n < a.se) ? swishf_(v + br_pre) : 0.0f;
  }
  __syncthreads();
  if (tid < CEXP) {
#pragma unroll
    for (int gi2 = 0; gi2 < G; ++gi2) {
      float v = be_pre;
#pragma unroll
      for (int n2 = 0; n2 < SE_MAX; ++n2) v += s_r[gi2 * 16 + n2] * we_pre[n2];
      const float gt = sigmoidf_(v);
      s_gate[gi2 * CEXP + tid] = gt;
      if (a.dbg_gate && gi2 < gvalid) a.dbg_gate[(size_t)(b0 + gi2) * CEXP + tid] = gt;
    }
  }
  __syncthreads();
  }
  ROWS_T(5);

  // ---- gated projection of this wave's row tile in each of its clips: D[j] * gate is the B fragment of chunk j, every weight fragment of the
  //      ring feeds CPW tiles ----
  if (!ROWS_ABL(32)) {
    f32x4 acc[CPW][NTP];
#pragma unroll
    for (int cw = 0; cw < CPW; ++cw)
#pragma unroll
      for (int nt = 0; nt < NTP; ++nt) acc[cw][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* gp = s_gate + gi0 * CEXP + 4 * g;
    // parked chunks come back through a two-deep register ring, requested two chunks ahead
    f32x4 pk[CPW][2];
#pragma unroll
    for (int cw = 0; cw < CPW; ++cw)
#pragma unroll
      for (int d = 0; d < 2; ++d)
        if (d < PARK) pk[cw][d] = *reinterpret_cast<const f32x4*>(park + (cw * PARK + d) * 256);
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      f32x4 bfrag[CPW];
#pragma unroll
      for (int cw = 0; cw < CPW; ++cw) {
        f32x4 dj;
        if (j < KR) dj = D[cw][j < KR ? j : 0];
        else {
          dj = pk[cw][(j - KR) & 1];
          if (j - KR + 2 < PARK) pk[cw][(j - KR) & 1] = *reinterpret_cast<const f32x4*>(park + (cw * PARK + j - KR + 2) * 256);
        }
        bfrag[cw] = dj * *reinterpret_cast<const f32x4*>(gp + cw * CEXP + 16 * j);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
          for (int cw = 0; cw < CPW; ++cw) acc[cw][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[j % PD][nt][s4], bfrag[cw][s4], acc[cw][nt], 0, 0, 0);
      if (j + PD < KC) {
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) wq[j % PD][nt] = p_w.ld((size_t)((j + PD) * NTP + nt) * 256);
      }
    }
#pragma unroll
    for (int cw = 0; cw < CPW; ++cw) {
      if (out_row && gi0 + cw < gvalid) {
        const size_t rin = (size_t)(b0 + gi0 + cw) * HW + q, rout = (size_t)(b0 + gi0 + cw) * HoWo + q;
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) {
          const int n = nt * 16 + 4 * g;
          if (n < a.Cout) {
            f32x4 y = acc[cw][nt] * *reinterpret_cast<const f32x4*>(a.scP + n) + *reinterpret_cast<const f32x4*>(a.shP + n);
            if (a.residual) y += *reinterpret_cast<const f32x4*>(a.X + rin * a.Cin + n);
            *reinterpret_cast<f32x4*>(a.Y + rout * a.Cout + n) = y;
          }
        }
      }
    }
  }
  ROWS_T(6);
  ROWS_T_STORE();
}

template <int KS, int S, int KCT, int HT, int WT, int CEXP, int NTP, int G, int CPW, int LDE, int SWZ, int PARK, int WPE>
int launch_rows_inst(hipStream_t s, const MidArgs& a) {
  using GM = RowsGeom<KS, S, KCT, HT, WT, CEXP, NTP, G, CPW, LDE, SWZ, PARK>;
  constexpr size_t lds = (size_t)GM::lds_floats * sizeof(float);
  auto* fn = &mbconv_rows_kernel<KS, S, KCT, HT, WT, CEXP, NTP, G, CPW, LDE, SWZ, PARK, WPE>;
  if (lds > 64 * 1024) {
    static bool raised[16] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(MKWS_ERR_HIP, "hipGetDevice failed");
    if (dev < 0 || dev >= 16 || !raised[dev]) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return fail(MKWS_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
      if (dev >= 0 && dev < 16) raised[dev] = true;
    }
  }
#ifdef MKWS_FRONT_TIMING
  static unsigned long long* d_t = nullptr;
  if (!d_t) (void)hipMalloc(&d_t, sizeof(unsigned long long) * 8 * 65536);
  MidArgs at = a; at.dbg_t = d_t;
  at.ablate = getenv("MKWS_ABLATE") ? atoi(getenv("MKWS_ABLATE")) : 0;
  const unsigned nb = (a.B + G - 1) / G;
  hipLaunchKernelGGL(fn, dim3(nb), dim3(GM::NTHR), lds, s, at);
  (void)hipStreamSynchronize(s);
  std::vector<unsigned long long> h((size_t)nb * 8);
  (void)hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost);
  double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < nb; ++i) for (int k = 0; k < 8; ++k) ph[k] += (double)h[8 * i + k];
  fprintf(stderr, "[rows-timing] KS %d H %d CEXP %d G %d CPW %d ablate %d: %u workgroups x %d thr, lds %zu | cycles per workgroup (thread 0): prologue %.0f  expand %.0f  barrier %.0f  "
          "stage+depthwise %.0f  presum %.0f  SE %.0f  project %.0f  | total %.0f\n", KS, HT, CEXP, G, CPW, at.ablate, nb, GM::NTHR, lds,
          ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[3] / nb, ph[4] / nb, ph[5] / nb, ph[6] / nb, ph[7] / nb);
#else
  hipLaunchKernelGGL(fn, dim3((a.B + G - 1) / G), dim3(GM::NTHR), lds, s, a);
#endif
  return MKWS_OK;
}

}  // namespace

const char* rows_kernel_name(int variant) {
  switch (variant) {
    case kRows2b: return "mbconv_rows_kernel<3,1,13,10,144,2x2>";
    case kRows3b: return "mbconv_rows_kernel<5,1,7,5,240,2x1>";
    case kRows2b + 16: return "mbconv_rows_kernel<3,1,13,10,144,1x1>";
    case kRows3b + 16: return "mbconv_rows_kernel<5,1,7,5,240,4x2>";
    default: return "mbconv_rows_kernel<?>";
  }
}

int launch_rows_variant(hipStream_t s, int variant, const MidArgs& a) {
  switch (variant) {
    // Workgroups per CU by registers = floor(waves per SIMD the register count allows / ceil(waves of a workgroup / 4)): a 9-wave workgroup
    // counts as 3 waves on EVERY SIMD, so two of them need 6 per SIMD = at most 80 registers; a 6-wave workgroup counts as 2: 128 registers.
    //                                     KS S KCT  H   W  CEXP NTP G CPW LDE SWZ PARK WPE
    case kRows2b: return launch_rows_inst<3, 1, 2, 13, 10, 144, 2, 2, 2, 20, 1, 0, 3>(s, a);   // 9 waves x 2 clips each, 81 KB of LDS, <= 168 registers: one workgroup per CU
    case kRows3b: return launch_rows_inst<5, 1, 3, 7, 5, 240, 3, 2, 1, 16, 0, 5, 4>(s, a);     // 6 waves (two clips), 78 KB, <= 128 registers: two workgroups per CU
    case kRows2b + 16: return launch_rows_inst<3, 1, 2, 13, 10, 144, 2, 1, 1, 20, 1, 4, 6>(s, a);   // (A/B: one clip per wave, 9 waves, 79 KB, <= 80 registers: two workgroups per CU)
    case kRows3b + 16: return launch_rows_inst<5, 1, 3, 7, 5, 240, 3, 4, 2, 16, 0, 0, 2>(s, a);     // (A/B: 6 waves x 2 clips each, <= 256 registers: one workgroup per CU)
    default: return fail(MKWS_ERR_INVALID_ARG, "unknown rows-kernel variant %d", variant);
  }
}

}  // namespace mkws

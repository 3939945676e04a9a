// 3-D multilinear on the fully overlapped brick table with the points of every wave ordered by
// their leading cell index on chip, and all waves of the chip walking that index in step with a
// clock ("sweep" evaluation; round 5).
//
// Why: on a 64^3 f64 grid the one-line-per-cell table (linear_brick.h, steps 1,1) is 10.2 MiB and
// unordered points miss the 4 MiB L2 of an XCD on 63 % of their lines; the kernel is then bound by
// the fabric's request rate.  Points ordered by the cell index of
// dimension 0 inside windows of W points make the workgroups of an XCD sweep the table once per
// window, and the table lines an XCD fetches per point drop to (lines of the table) / (W / 8):
// profiles/r05_slab_window_*.jsonl (the brick kernel on window-ordered points) has 1.44 ms per 1e8
// points unordered, 1.01 ms at W = 2^21, 0.97 at 2^22, 0.91 fully ordered.  W is what the chip can
// hold at a time, and the points must be back in their own order when the results are stored.
//
// How:
//  * every wave keeps K x 64 points in REGISTERS (the LDS alone would hold 1.7e6 points chip-wide
//    at 24 bytes each, the register files three times that), counting-sorts them by leading cell
//    index through LDS — per wave: there is no workgroup barrier and no wait on another wave
//    anywhere —, evaluates its K rows in that order with linear_brick.h's quad-cooperative gather,
//    and puts the results back into the points' own order through LDS, so that coordinates are
//    read and results written exactly as coalesced as in the brick kernel;
//  * "row k at the same time on every CU of an XCD" cannot come from when waves start (they drift).
//    It comes from a clock every wave can read, the constant-rate counter s_memrealtime: one sweep
//    of the leading index takes `period` ticks; a wave that begins its rows at time t begins with
//    the row (a quantile of its sorted points) that the clock's phase frac(t / period) names and
//    wraps around.  All waves that gather at a given moment then gather from the same slab of the
//    table, whenever they started (stamps: tools/sweep_clock_probe.py).  The period that works is
//    the mean duration of a round; every launch measures it and leaves it for the next one
//    (SweepWork::period; the first launch of a handle uses a default);
//  * rounds are dealt on demand from eight counters (one per XCD, a wave steals from the others'
//    once its own is empty): with a static deal the slowest waves ended 12 % after the mean.
// A point's result depends on its coordinates only (src/multilinear/regular.rs:276-280) and the
// arithmetic per point is linear_brick.h's (the reference's operations in its order,
// regular.rs:296-404): results are bit-identical whatever the order, the period or the deal.
#pragma once

#include "linear_brick.h"
#include "sweep_rounds.h"  // SweepWork; the scaffold the other sweep kernels share (this kernel keeps its own copy: see below)

namespace interpn {

// (SweepWork — the work words of the launches through one scratch block — lives in sweep_rounds.h)

template <typename T>
struct SweepArgs {
  BrickArgs<T, 3> b;   // bricks (steps SI, SJ), obs, out, first_bad, npts, start, step, n, nbj, nbk
  T key_start, key_scale;  // the sort key: (x0 - key_start) * key_scale ~ the leading cell index (a locality hint: it need not be the exact cell; rectilinear: the uniform grid over the axis' span)
  int key_shift;       // leading cell index >> key_shift < 64 bins
  T rstep[3];          // regular grids: RN(1 / step[d]), computed by the host in T (interpn_device.h: step_cell_fast)
  unsigned fastdiv;    // != 0: every step is in the range where step_cell_fast is the reference's value (2^-128 <= |step| <= 2^128, finite)
  unsigned rounds;     // 64 * K points each
  unsigned per_shard;  // rounds per shard (8 shards)
  unsigned period;     // > 0: ticks per sweep, overriding the measured one; 1: rows in sorted order (no clock)
  unsigned period_default;  // before anything has been measured
  unsigned gated;           // != 0: do nothing if work->take_brick is set (see SweepWork, sweep_rounds.h)
  SweepWork* work;
  unsigned long long* stamps;  // STAMPS builds (tools/): 8 words per wave, see the kernel's end
};

// K rows of 64 points per wave and round; THREADS per workgroup.
// LDS per wave: a row buffer of K * 512 bytes (sort exchange of the K rows that go back to registers;
// the KL parked rows are written by the exchange straight to where they wait) that shares its bytes with
// the quad-transposition pieces of linear_brick.h (5 KiB, used by the rows in between); behind it the
// parked rows' coordinates; the result exchange after the rows spans both (every parked row has been
// evaluated by then); + 1 KiB of piece offsets, whose bytes are the sort's 64 counters x 2 before the rows.
template <typename T, int K, int KL = 0>
struct SweepLds {
  static constexpr unsigned kRowOnly = 64u * K * sizeof(T);
  static constexpr unsigned kPiece = 64u * kPieceRow * 2u * sizeof(T);
  static constexpr unsigned kRow = kRowOnly > kPiece ? kRowOnly : kPiece;
  static constexpr unsigned kPark = 64u * KL * 3u * sizeof(T);   // the KL rows whose coordinates wait in LDS instead of registers: [3][KL * 64] by sorted position
  static constexpr unsigned kOff = 64u * 16u;                    // piece offsets of a row (during the rows) | points per bin, first position per bin (during the sort)
  static constexpr unsigned kWave = kRow + kPark + kOff;
  static constexpr unsigned kWorkgroup = 16;  // behind the waves' regions: ticks | rounds | waves done | -
  static_assert(kRow + kPark >= 64u * (K + KL) * sizeof(T), "the result exchange spans the row buffer and the parked rows' bytes");
  static_assert(64u * 4u * 2u <= kOff, "the sort's counters live in the piece offsets' bytes");
  static_assert(64u * (K + KL) * 2u <= kRow, "the 16-bit source-index exchange of a round stays inside the row buffer (the parked coordinates sit right behind it)");
};

// RECT: rectilinear grids, the cell search exactly as in the brick kernel — axes of at most 64
// coordinates held one coordinate per lane and searched with cross-lane reads (lane_axes.h; AXR =
// its mode 1..3); longer axes (AXR = 4) searched with interpn_device.h::axis_cell in the axis image
// (coordinates + bucket tables, or per-bucket records) that the workgroup stages into LDS behind
// its waves' regions (a.ax.use_lds), or through L1/L2 where that image is too large.
// CELL: the brick form (linear_brick.h) — 0: 2 x 2 x KW bricks stepped (SI, SJ); 2 (f32): 2 x 4 x 4 bricks, one line per cell.
// KL: rows per wave and round BESIDE the K in registers, their coordinates parked in LDS between the sort and their
//     turn (the kernel leaves 68 KiB of a CU's LDS unused at K = 12; the window is what bounds the table misses:
//     table lines x 8 / points held chip-wide).
// ABL: measurement builds (tools/ablate_linear3d.hip only; 0 in the library) — 1: no table access (cell values made up
//     from the coordinates), 2: no streams (coordinates made up from the point index, nothing stored), 3: the six IEEE divisions per point as multiplications by a reciprocal (timing only).
template <typename T, bool RECT, bool FMA, int SI, int SJ, int K, int THREADS, int AXR = 0, bool STAMPS = false, int CELL = 0, int KL = 0, int ABL = 0>
__global__ void __launch_bounds__(THREADS) k_linear_sweep(const SweepArgs<T> s) {
  constexpr int PPV = 16 / (int)sizeof(T);  // points per 16-byte stream access: 2 (f64) or 4 (f32)
  constexpr int KT = K + KL;                // rows per wave and round
  static_assert(KT % PPV == 0 && KT % 2 == 0 && K >= PPV && KT <= 32 && KL >= 0, "rows per wave and round");
  static_assert(CELL == 0 || (CELL == 2 && sizeof(T) == 4 && SI == 1 && SJ == 1), "2 x 4 x 4 bricks: f32");
  static_assert(RECT == (AXR != 0), "rectilinear grids: lane-resident axes only");
  typedef typename LeafVec<T, 2>::type P;
  typedef T TV __attribute__((ext_vector_type(PPV)));
  typedef BrickGeom<T, CELL> Geom;
  constexpr int SK = Geom::SK;
  typedef SweepLds<T, K, KL> L;
  const BrickArgs<T, 3>& a = s.b;
  if (s.gated && __hip_atomic_load(&s.work->take_brick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // (launch-uniform: no work word is touched)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = threadIdx.x >> 6;
  unsigned char* const mine = smem_raw + wave * L::kWave;
  T* const row = reinterpret_cast<T*>(mine);                                 // [64 K]
  typedef unsigned short __attribute__((may_alias)) lds_u16;
  lds_u16* const row16 = reinterpret_cast<lds_u16*>(mine);                   // [64 K] (the row buffer's bytes, between its uses)
  P* const lds_piece = reinterpret_cast<P*>(mine);                           // [16 quads][4][kPieceRow] (the same bytes, during the rows)
  T* const park = reinterpret_cast<T*>(mine + L::kRow);                      // [3][KL * 64] coordinates of the rows beyond the registers, by sorted position
  lds_u32* const lds_off = reinterpret_cast<lds_u32*>(mine + L::kRow + L::kPark);
  lds_u32* const cnt = lds_off;                                              // [64] points per bin, then first position per bin at [64..128): the same bytes, before the rows
  lds_u32* const wg_words = reinterpret_cast<lds_u32*>(smem_raw + (THREADS / 64) * L::kWave);  // ticks | rounds | waves done
  if (threadIdx.x < 4) wg_words[threadIdx.x] = 0;
  if constexpr (RECT && AXR == 4) {
    if (s.b.ax.use_lds) {
      const unsigned words = s.b.ax.image_bytes >> 2;
      const unsigned* src = reinterpret_cast<const unsigned*>(s.b.ax.image);
      unsigned* dst = reinterpret_cast<unsigned*>(smem_raw + (THREADS / 64) * L::kWave + L::kWorkgroup);
      for (unsigned k = threadIdx.x; k < words; k += THREADS) dst[k] = src[k];
    }
  }
  __syncthreads();  // the only workgroup barrier: before any wave has taken work
  const unsigned q = lane & 3u;
  const unsigned quad = lane >> 2;
  constexpr size_t kChunk = (size_t)64 * KT;
  const unsigned nwaves = gridDim.x * (THREADS / 64);
  LaneAxes<T, 3> la;
  if constexpr (RECT && AXR <= 3) la = load_lane_axes<T, 3, AXR>(a.ax);
  const unsigned char* axis_base = a.ax.image;
  if constexpr (RECT && AXR == 4) {
    if (a.ax.use_lds) {  // staged before the workgroup's one barrier (above: wg_words)
      axis_base = smem_raw + (THREADS / 64) * L::kWave + L::kWorkgroup;
    }
  }
  SweepWork* const work = s.work;
  unsigned period = s.period;
  if (period == 0) {
    period = __hip_atomic_load(&work->period, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (period == 0) period = s.period_default;
  }
  period = __builtin_amdgcn_readfirstlane(period);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned shard = xcc & 7u;
  // One returning atomic per round and wave, eight counters: 1e8 points are 1.3e5..2e5 rounds per
  // millisecond, and one word serves ~9e4 returning atomics per millisecond.
  auto take = [&](unsigned sh) -> unsigned {
    unsigned v = 0;
    if (lane == 0) v = atomicAdd(&work->head[sh][0], 1u);
    return v;  // lane 0 holds the answer; consumed through readfirstlane
  };
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_mark = t_begin, st_other = 0, st_rows = 0;
  unsigned st_rot = 0;
  unsigned my_rounds = 0;
  __builtin_amdgcn_s_setprio(2);  // (see the rows below)
  unsigned ticket = take(shard);
  while (true) {
    unsigned rr = __builtin_amdgcn_readfirstlane(ticket);
    if (rr >= s.per_shard || shard * s.per_shard + rr >= s.rounds) {  // this shard is empty: the next one that is not
      // (a look before the returning atomic: at the end of a launch every wave comes through here,
      // and thousands of read-modify-writes on eight words took 50-100 us; loads do not queue up)
      bool found = false;
      for (unsigned c = 1; c < 8 && !found; ++c) {
        const unsigned sh = (shard + c) & 7u;
        const unsigned seen = __hip_atomic_load(&work->head[sh][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen < s.per_shard && sh * s.per_shard + seen < s.rounds) { found = true; shard = sh; }
      }
      if (!found) break;  // (the counters only grow: a wave that loses the race for a shard's last rounds looks again and ends here)
      ticket = take(shard);
      continue;
    }
    const unsigned r = shard * s.per_shard + rr;
    ticket = take(shard);  // the next round's ticket travels while this round's coordinates do
    ++my_rounds;
    const size_t base = (size_t)r * kChunk;
    // -- coordinates: K / PPV 16-byte loads per dimension, lane l holds points kv * 64 PPV + PPV l + {0 .. PPV - 1}
    T x[KT][3];
    const bool full = base + kChunk <= a.npts;
    if (full) {  // (all but the batch's last round: the loads back to back, nothing between them)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int kv = 0; kv < KT / PPV; ++kv) {
          TV v;
          if constexpr (ABL == 2) {
#pragma unroll
            for (int h = 0; h < PPV; ++h) v[h] = ablate_coord<T>(base + (size_t)(kv * 64 + (int)lane) * PPV + h, d, a.start[d], a.step[d], a.n[d]);
          } else {
            v = stream_load(reinterpret_cast<const TV*>(a.obs[d] + base) + (kv * 64 + (int)lane));
          }
#pragma unroll
          for (int h = 0; h < PPV; ++h) x[PPV * kv + h][d] = v[h];
        }
    } else {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int kv = 0; kv < KT / PPV; ++kv) {
          const size_t i0 = base + (size_t)kv * (64 * PPV) + PPV * lane;
          TV v;
#pragma unroll
          for (int h = 0; h < PPV; ++h) v[h] = a.start[d];
          if (i0 + PPV - 1 < a.npts) {
            v = stream_load(reinterpret_cast<const TV*>(a.obs[d] + i0));
          } else {
#pragma unroll
            for (int h = 0; h < PPV; ++h)
              if (i0 + h < a.npts) v[h] = stream_load(a.obs[d] + i0 + h);
          }
#pragma unroll
          for (int h = 0; h < PPV; ++h) x[PPV * kv + h][d] = v[h];
        }
    }
    // -- counting sort of the wave's 64 K points by leading cell index (a hint: NaN -> bin 0)
    cnt[lane] = 0;
    wave_sync();
    unsigned pos[KT];  // rank inside (wave, bin) | bin << 16, then the sorted position
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const T u = (x[k][0] - s.key_start) * s.key_scale;
      int c = u >= (T)1 ? (u < (T)(a.n[0] - 2) ? (int)u : a.n[0] - 2) : 0;
      const unsigned bin = (unsigned)(c >> s.key_shift);
      pos[k] = atomicAdd(&cnt[bin], 1u) | (bin << 16);
    }
    wave_sync();
    {
      const unsigned mine_cnt = cnt[lane];
      unsigned incl = mine_cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, off);
        if (lane >= (unsigned)off) incl += up;
      }
      cnt[64 + lane] = incl - mine_cnt;
    }
    wave_sync();
    // the first row: the one the clock's phase names (see the head of this file)
    unsigned rot = 0;
    if (period > 1) {
      const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
      const unsigned ph = now % period;
      rot = __builtin_amdgcn_readfirstlane((unsigned)(((unsigned long long)ph * KT) / period) * 64u);
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      unsigned p = (pos[k] & 0xFFFFu) + cnt[64 + (pos[k] >> 16)];
      p = p >= rot ? p - rot : p + (unsigned)(64 * KT) - rot;
      pos[k] = p;
    }
    // (sorted positions K * 64 and up are the parked rows: their coordinates go straight to where they wait,
    //  park[d][position - K * 64], which lies kParkSkip + d * KL * 64 elements behind the same index of `row`)
    constexpr unsigned kParkSkip = L::kRow / sizeof(T) - 64u * K;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        unsigned at = pos[k];
        if constexpr (KL > 0) at += pos[k] >= 64u * K ? kParkSkip + (unsigned)d * (64u * KL) : 0u;
        row[at] = x[k][d];
      }
      wave_sync();
#pragma unroll
      for (int k = 0; k < K; ++k) x[k][d] = row[k * 64 + lane];
      wave_sync();
    }
    unsigned src[KT / 2];  // where (inside the chunk) the points in my slots 2 k2 | 2 k2 + 1 came from (16 bits each)
#pragma unroll
    for (int k = 0; k < KT; ++k) row16[pos[k]] = (unsigned short)((k / PPV) * (64 * PPV) + PPV * lane + (k % PPV));
    wave_sync();
#pragma unroll
    for (int k2 = 0; k2 < KT / 2; ++k2) src[k2] = (unsigned)row16[(2 * k2) * 64 + lane] | ((unsigned)row16[(2 * k2 + 1) * 64 + lane] << 16);
    wave_sync();
    // -- K rows in sorted order
    if constexpr (STAMPS) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); st_other += now - st_mark; st_mark = now; st_rot = rot; }
    // Wave priority (round 6): the rows — a chain of gathers the wave mostly waits in — run at priority 0, everything around
    // them (result exchange and stores, ticket, coordinate loads, sort) at 2: a wave that has finished its rows gets its
    // stores out and its next round's loads under way ahead of the SIMD's other waves' row arithmetic (cfg2 -1 %, cfg3 -2 %,
    // the 128^3 shard -1 %: tools/ab_sweep.py; priority during the loads alone, or raised during the rows: nothing).
    __builtin_amdgcn_s_setprio(0);
    T res[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      // one row at a time ...
      __builtin_amdgcn_sched_barrier(0);
      T xr[3];  // this row's point: from the registers, or from where the sort parked it
#pragma unroll
      for (int d = 0; d < 3; ++d) xr[d] = k < K ? x[k < K ? k : 0][d] : park[(d * KL + (k - K)) * 64 + lane];
      // rectilinear grids: t's divisions behind the gather's loads (f32 64^3 0.65 -> 0.63 ms) — except f64 with the axes in
      // lanes, where the six more registers live across the loads spill (cfg3 +2.5 %: tools/ab_sweep.py)
#ifndef INTERPN_SWEEP_LANES_LATE
#define INTERPN_SWEEP_LANES_LATE 0
#endif
      constexpr bool RECT_T_LATE = RECT && (sizeof(T) == 4 || AXR == 4 || INTERPN_SWEEP_LANES_LATE);
      T t[3];
      int loc[3];
      T av[3] = {(T)0, (T)0, (T)0};  // the dividends of t (regular: x - izl; rectilinear: x - x0), the quotients taken behind the gather's loads
      T bv[3] = {(T)1, (T)1, (T)1};  // rectilinear: the divisors x1 - x0
      bool have_t = false;          // ... unless the wave took the divide sequences
      if constexpr (RECT && AXR == 4) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const Axis<T> ax = make_axis<T, 3>(a.ax, axis_base, d);
          T x0, x1;
          loc[d] = axis_cell<T>(ax, xr[d], &x0, &x1);   // multilinear/rectilinear.rs:353-370, :310-311
          av[d] = xr[d] - x0;                           // rectilinear.rs:310-313: t = (x - x0) / (x1 - x0), the division
          bv[d] = x1 - x0;                              // behind the gather's loads
        }
      } else if constexpr (RECT) {
        T xin[1][3], x0_r[1][3], x1_r[1][3];
        int cell_r[1][3];
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[0][d] = xr[d];
        lane_axes_locate<T, 3, 1, AXR>(a.ax, la, xin, cell_r, x0_r, x1_r);  // multilinear/rectilinear.rs:353-370
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          av[d] = xr[d] - x0_r[0][d];            // rectilinear.rs:310-313: t = (x - x0) / (x1 - x0), the division
          bv[d] = x1_r[0][d] - x0_r[0][d];       // behind the gather's loads where the registers allow (RECT_T_LATE)
          if constexpr (!RECT_T_LATE) t[d] = av[d] / bv[d];
          loc[d] = cell_r[0][d];
        }
      } else {
        bool ok = true;
        bool exact = false;
        if constexpr (ABL != 3) {
          // cell index and t without the six divide sequences (interpn_device.h::step_cell_fast: the same bits)
          exact = s.fastdiv != 0;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            // (step_cell_fast in two halves: the cell here, in front of the gather's loads; the quotient behind them)
            const StepCellIndex<T> sc = step_cell_index<FMA>(xr[d], a.start[d], a.step[d], s.rstep[d], a.n[d] - 2);
            av[d] = sc.a;
            loc[d] = sc.loc;
            exact = exact && sc.exact;
          }
        }
        if (__any(!exact)) {
          have_t = true;  // a lane on a grid plane, far outside the grid, not finite ...: the reference's operations as they stand, for the wave
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            T floc;
            if constexpr (ABL == 3) {  // (timing only: reciprocal multiplications, results differ in the last bits)
              const T rs = (T)1 / a.step[d];
              floc = dev_floor<T>((xr[d] - a.start[d]) * rs);
              const int l = clamp_loc<T>(floc, a.n[d] - 2);
              const T izl = mul_add<FMA>(a.step[d], (T)l, a.start[d]);
              t[d] = (xr[d] - izl) * rs;
              loc[d] = l;
              continue;
            }
            ok &= regular_floc<T>(xr[d], a.start[d], a.step[d], &floc);  // multilinear/regular.rs:415-418
            const int l = clamp_loc<T>(floc, a.n[d] - 2);                  // regular.rs:420-422
            const T izl = mul_add<FMA>(a.step[d], (T)l, a.start[d]);       // regular.rs:334-337
            t[d] = (xr[d] - izl) / a.step[d];                            // regular.rs:339
            loc[d] = l;
          }
        }
        const size_t gi = base + ((src[k / 2] >> (16 * (k & 1))) & 0xFFFFu);
        if (!ok && gi < a.npts) atomicMin(a.first_bad, (unsigned long long)gi);
      }
      const unsigned bk = (unsigned)loc[2] / (unsigned)SK;
      Cell<T> c;
      if constexpr (SI == 1 && SJ == 1 && ABL != 1) {
        // One line per cell: the point's four pieces lie at fixed distances from `mine_b`, the byte offset of
        // its first one (2 x 2 x KW bricks: KW elements apart; the f32 2 x 4 x 4 bricks: rows (di, oj + dj) of four).
        // Lane q of a quad loads piece q of the quad's points 0..3: their offsets come across the quad by DPP
        // (no LDS round trip in the row's dependent chain), and 32-bit byte offsets beside the table's base
        // keep the address arithmetic off the vector unit (the host takes this kernel for tables under 4 GiB only).
        const unsigned mine_b = brick_line_bytes24<T, CELL>(a.nbj, a.nbk, loc[0], loc[1], loc[2]);  // (full-rate integer arithmetic: linear_brick.h)
        const unsigned char* const tb = reinterpret_cast<const unsigned char*>(a.bricks);
        const unsigned mypiece = (CELL == 2 ? ((q >> 1) * 4u + (q & 1u)) * 4u : q * (unsigned)Geom::KW) * (unsigned)sizeof(T);
        P pc[4];
        pc[0] = *reinterpret_cast<const P*>(tb + ((unsigned)__builtin_amdgcn_mov_dpp((int)mine_b, 0x00, 0xF, 0xF, true) + mypiece));
        pc[1] = *reinterpret_cast<const P*>(tb + ((unsigned)__builtin_amdgcn_mov_dpp((int)mine_b, 0x55, 0xF, 0xF, true) + mypiece));
        pc[2] = *reinterpret_cast<const P*>(tb + ((unsigned)__builtin_amdgcn_mov_dpp((int)mine_b, 0xAA, 0xF, 0xF, true) + mypiece));
        pc[3] = *reinterpret_cast<const P*>(tb + ((unsigned)__builtin_amdgcn_mov_dpp((int)mine_b, 0xFF, 0xF, 0xF, true) + mypiece));
        if constexpr (RECT && RECT_T_LATE) {  // t: only the lerps need it — the three divide sequences under the loads' latency
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int d = 0; d < 3; ++d) t[d] = av[d] / bv[d];
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!RECT && ABL != 3) {  // t: only the lerps need it — under the loads' latency (cfg2 -0.5 %)
          __builtin_amdgcn_sched_barrier(0);
          if (!have_t) {
#pragma unroll
            for (int d = 0; d < 3; ++d) t[d] = step_cell_quotient(av[d], a.step[d], s.rstep[d]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) lds_piece[(quad * 4 + r) * kPieceRow + q] = pc[r];
        wave_sync();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const P w = lds_piece[(quad * 4 + q) * kPieceRow + p];
          c.v[p >> 1][p & 1][0] = w.x;
          c.v[p >> 1][p & 1][1] = w.y;
        }
        wave_sync();
      } else {
        if constexpr (RECT && RECT_T_LATE) {
#pragma unroll
          for (int d = 0; d < 3; ++d) t[d] = av[d] / bv[d];
        }
        if constexpr (!RECT && ABL != 3) {  // (the other layouts and the no-table measurement build: the quotients in front)
          if (!have_t) {
#pragma unroll
            for (int d = 0; d < 3; ++d) t[d] = step_cell_quotient(av[d], a.step[d], s.rstep[d]);
          }
        }
        const unsigned kpart = bk * (unsigned)Geom::ELEMS + ((unsigned)loc[2] - bk * (unsigned)SK);
#pragma unroll
        for (int p = 0; p < 4; ++p)
          lds_off[(quad * 4 + p) * 4 + q] = brick_piece<T, SI, SJ, CELL>(a.nbj, a.nbk, loc[0], loc[1], kpart, p >> 1, p & 1);
        wave_sync();
        const uint4 toff = *reinterpret_cast<const uint4*>(&lds_off[(quad * 4 + q) * 4]);
        if constexpr (ABL == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) c.v[e >> 2][(e >> 1) & 1][e & 1] = t[e % 3] + (T)(toff.x & 7u) + (T)e;
        } else {
          c = gather_cell<T>(a.bricks, toff, 0u, lds_piece, quad, q);
        }
      }
      T rr2[2];
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {  // reference order (regular.rs:347-403): i first, k last
        const T c0 = mul_add<FMA>(t[0], c.v[1][0][dk] - c.v[0][0][dk], c.v[0][0][dk]);
        const T c1 = mul_add<FMA>(t[0], c.v[1][1][dk] - c.v[0][1][dk], c.v[0][1][dk]);
        rr2[dk] = mul_add<FMA>(t[1], c1 - c0, c0);
      }
      res[k] = mul_add<FMA>(t[2], rr2[1] - rr2[0], rr2[0]);
      // ... and finished before the next begins: the optimiser otherwise sinks a row's divisions and
      // lerps down to the result exchange and keeps its eight cell values alive (K = 8: 146 spilled
      // registers at 128; with the two pins: none)
      asm volatile("" : "+v"(res[k]));
    }
    if constexpr (STAMPS) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); st_rows += now - st_mark; st_mark = now; }
    __builtin_amdgcn_s_setprio(2);
    // -- results back into the points' own order through LDS
#pragma unroll
    for (int k = 0; k < KT; ++k) row[(src[k / 2] >> (16 * (k & 1))) & 0xFFFFu] = res[k];
    wave_sync();
#pragma unroll
    for (int kv = 0; kv < KT / PPV; ++kv) {
      const size_t i0 = base + (size_t)kv * (64 * PPV) + PPV * lane;
      const TV v = *reinterpret_cast<const TV*>(&row[kv * (64 * PPV) + PPV * lane]);
      if constexpr (ABL == 2) {
        if (v[0] == (T)123.456) stream_store(reinterpret_cast<TV*>(a.out + i0), v);  // (never)
      } else if (full || i0 + PPV - 1 < a.npts) {
        stream_store(reinterpret_cast<TV*>(a.out + i0), v);
      } else {
#pragma unroll
        for (int h = 0; h < PPV; ++h)
          if (i0 + h < a.npts) stream_store(a.out + i0 + h, v[h]);
      }
    }
    wave_sync();
  }
  // -- this wave is done: its share of the period measurement goes to the workgroup's LDS words; the
  //    workgroup's last wave adds them to the launch's, and the launch's last workgroup turns the
  //    sums into the next launch's period and leaves the work words zero.  (Every wave adding to the
  //    launch's words itself: 3 x 3072 read-modify-writes on two lines = 20-65 us at the end of
  //    every launch, stamps of round 5.)
  const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const unsigned my_ticks = (unsigned)(t_end - t_begin);  // < 43 s
    atomicAdd(&wg_words[0], my_ticks);   // a workgroup's waves together stay below 2^32 ticks for launches under 2.6 s
    atomicAdd(&wg_words[1], my_rounds);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (atomicAdd(&wg_words[2], 1u) == THREADS / 64 - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const unsigned wg_ticks = wg_words[0], wg_rounds = wg_words[1];
      if (wg_rounds) {  // returning forms, answers consumed: both have been performed before `done` is touched
        const unsigned long long r1 = atomicAdd(&work->ticks, (unsigned long long)wg_ticks);
        const unsigned r2 = atomicAdd(&work->rounds, wg_rounds);
        asm volatile("" ::"v"(r1), "v"(r2));
      }
      const unsigned d = atomicAdd(&work->done[0], 1u);
      if (d == gridDim.x - 1) {
        const unsigned long long ticks = atomicAdd(&work->ticks, 0ull);
        const unsigned rounds = atomicAdd(&work->rounds, 0u);
        // a launch of few rounds per wave measures its start-up, not a round: keep what there was
        if (rounds >= 4 * nwaves) {
          // 0.9 x the mean round: measured optimum (a period a little short of the round spreads the
          // waves of an XCD over a few rows instead of one, i.e. over more L2 channels; fed back as it
          // is the mean settles at a longer, slower period)
          unsigned long long p = ticks * 9 / ((unsigned long long)rounds * 10);
          p = p < 200 ? 200 : (p > 20000 ? 20000 : p);  // 2 .. 200 us
          atomicExch(&work->period, (unsigned)p);
        }
        for (int x = 0; x < 8; ++x) atomicExch(&work->head[x][0], 0u);
        atomicExch(&work->ticks, 0ull);
        atomicExch(&work->rounds, 0u);
        atomicExch(&work->done[0], 0u);
      }
    }
    if constexpr (STAMPS) {  // start | end | ticks outside the rows | ticks in rows | XCD | last first row | rounds | period
      if (s.stamps) {
        unsigned long long* o = s.stamps + (size_t)(blockIdx.x * (THREADS / 64) + wave) * 8;
        o[0] = t_begin; o[1] = t_end; o[2] = st_other; o[3] = st_rows;
        o[4] = xcc & 0xFu; o[5] = st_rot / 64u; o[6] = my_rounds; o[7] = period;
      }
    }
  }
}

}  // namespace interpn

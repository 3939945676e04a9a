// 3-D multicubic on the fully overlapped tile table with the points of every wave ordered on chip by
// their cell index along the table's slowest dimension, and all waves of the chip walking that index in
// step with a clock: the sweep evaluation of linear_sweep.h (read its head first: the window argument,
// the per-wave sort, the clock, the rounds dealt on demand are the same) with cubic_brick.h's rows.
//
// Why: a 3-D cubic point reads four tile lines (one per plane of its footprint along dim 2); on a 64^3
// f64 grid the one-tile-per-footprint table is 30 MiB and unordered points miss the L2 on every one of
// them: 5.1 GB over the fabric per 1e7 points, 0.60-0.65 ms.  The table is stored
// plane by plane (dim 2 slowest: cubic_brick.h), so points ordered by their dim-2 cell inside the window
// the chip holds make an XCD fetch every plane once per window and read it from its L2 four times per
// point.  The column evaluation of sorted points (cubic3_column.h) needs a global sort that costs more
// than it saves at this size; here the order is made on chip, in the pass that evaluates.
//
// The rows are cubic_brick.h's, instruction for instruction: per-dimension cell, saturation class and t
// (multicubic/regular.rs:435-466, rectilinear.rs:366-408), the plane's 64 tiles by LDS-DMA into the
// wave's image, the reference's reduction tree (regular.rs:368-421).  A point's result depends on its
// coordinates only: the same bits whatever the order, the period or the deal.
#pragma once

#include "cubic_brick.h"
#include <type_traits>

#include "linear_sweep.h"  // SweepWork (the launches' work words)

namespace interpn {

// f(integral_constant<int, I>) for I = FROM .. TO - 1 (a row's index must be a constant: its coordinates live in registers)
template <int FROM, int TO, typename F>
__device__ __forceinline__ void sweep_static_for(F&& f) {
  if constexpr (FROM < TO) {
    f(std::integral_constant<int, FROM>{});
    sweep_static_for<FROM + 1, TO>(f);
  }
}

template <typename T>
struct CubicSweepArgs {
  CubicBrickArgs<T, 3> c;  // bricks (steps 1,1), table_bytes, obs, out, first_bad, npts, start, step, n, ax, plane_stride, nbj, linearize
  T key_start, key_scale;  // the sort key: (x2 - key_start) * key_scale ~ the dim-2 cell index (a locality hint; rectilinear: the uniform grid over the axis' span)
  int key_shift;           // cell index >> key_shift < 64 bins
  unsigned rounds;         // 64 * (K + KL) points each
  unsigned per_shard;      // rounds per shard (8 shards)
  unsigned period;         // > 0: ticks per sweep, overriding the measured one; 1: rows in sorted order (no clock)
  unsigned period_default;
  T rstep[3];              // regular grids: RN(1 / step[d]), a division in T on the host (interpn_device.h: floor_quotient_fast / divide_fast)
  unsigned fastdiv;        // != 0: every step lies where those forms are the reference's values (StepCellRange<T>)
  SweepWork* work;
};

// LDS per wave: the tile image of cubic_brick.h's LDS-DMA gather (64 tiles: 8 KiB f64, 4 KiB f32), whose bytes are
// the sort's and the results' exchange buffer before and after the rows; the parked rows; the sort's counters.
template <typename T, int K, int KL>
struct CubicSweepLds {
  static constexpr unsigned kRowOnly = 64u * K * sizeof(T);
  static constexpr unsigned kImage = 64u * (unsigned)sizeof(T) * 16u;
  static constexpr unsigned kRow = kRowOnly > kImage ? kRowOnly : kImage;
  static constexpr unsigned kPark = 64u * KL * 3u * sizeof(T);
  static constexpr unsigned kCnt = 64u * 4u * 2u;
  static constexpr unsigned kWave = kRow + kPark + kCnt;
  static constexpr unsigned kWorkgroup = 16;
  static_assert(kRow + kPark >= 64u * (K + KL) * sizeof(T), "the result exchange spans the image and the parked rows' bytes");
};

template <typename T, bool RECT, bool FMA, int K, int KL, int THREADS>
__global__ void __launch_bounds__(THREADS) k_cubic_sweep(const CubicSweepArgs<T> s) {
  constexpr int PPV = 16 / (int)sizeof(T);
  constexpr int KT = K + KL;
  static_assert(KT % PPV == 0 && KT % 2 == 0 && K >= PPV && KT <= 32, "rows per wave and round");
  typedef T TV __attribute__((ext_vector_type(PPV)));
  typedef typename CubicDimSel<T, RECT>::type DimT;
  typedef CubicSweepLds<T, K, KL> L;
  const CubicBrickArgs<T, 3>& a = s.c;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = threadIdx.x >> 6;
  unsigned char* const mine = smem_raw + wave * L::kWave;
  T* const row = reinterpret_cast<T*>(mine);
  typedef unsigned short __attribute__((may_alias)) lds_u16;
  lds_u16* const row16 = reinterpret_cast<lds_u16*>(mine);
  T* const park = reinterpret_cast<T*>(mine + L::kRow);
  lds_u32* const cnt = reinterpret_cast<lds_u32*>(mine + L::kRow + L::kPark);
  lds_u32* const wg_words = reinterpret_cast<lds_u32*>(smem_raw + (THREADS / 64) * L::kWave);
  if (threadIdx.x < 4) wg_words[threadIdx.x] = 0;
  unsigned char* const lds_axes = smem_raw + (THREADS / 64) * L::kWave + L::kWorkgroup;
  if constexpr (RECT) {
    if (a.ax.use_lds) {
      const unsigned words = a.ax.image_bytes >> 2;
      const unsigned* src = reinterpret_cast<const unsigned*>(a.ax.image);
      unsigned* dst = reinterpret_cast<unsigned*>(lds_axes);
      for (unsigned k = threadIdx.x; k < words; k += THREADS) dst[k] = src[k];
    }
  }
  __syncthreads();  // the only workgroup barrier: before any wave has taken work
  const unsigned char* axis_base = (RECT && a.ax.use_lds) ? lds_axes : a.ax.image;
  const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(a.bricks, a.table_bytes);
  // LDS byte address of this wave's tile image, in a scalar register
  const unsigned lds_wave = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw + wave * L::kWave));
  constexpr size_t kChunk = (size_t)64 * KT;
  const unsigned nwaves = gridDim.x * (THREADS / 64);
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
  auto take = [&](unsigned sh) -> unsigned {
    unsigned v = 0;
    if (lane == 0) v = atomicAdd(&work->head[sh][0], 1u);
    return v;
  };
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  unsigned my_rounds = 0;
  unsigned ticket = take(shard);
  while (true) {
    unsigned rr = __builtin_amdgcn_readfirstlane(ticket);
    if (rr >= s.per_shard || shard * s.per_shard + rr >= s.rounds) {  // this shard is empty: the next one that is not
      bool found = false;
      for (unsigned c = 1; c < 8 && !found; ++c) {
        const unsigned sh = (shard + c) & 7u;
        const unsigned seen = __hip_atomic_load(&work->head[sh][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen < s.per_shard && sh * s.per_shard + seen < s.rounds) { found = true; shard = sh; }
      }
      if (!found) break;
      ticket = take(shard);
      continue;
    }
    const unsigned r = shard * s.per_shard + rr;
    ticket = take(shard);  // the next round's ticket travels while this round's coordinates do
    ++my_rounds;
    const size_t base = (size_t)r * kChunk;
    // -- coordinates
    T x[KT][3];
    const bool full = base + kChunk <= a.npts;
    if (full) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int kv = 0; kv < KT / PPV; ++kv) {
          const TV v = stream_load(reinterpret_cast<const TV*>(a.obs[d] + base) + (kv * 64 + (int)lane));
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
          for (int h = 0; h < PPV; ++h) v[h] = RECT ? (T)0 : a.start[d];
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
    // -- counting sort of the wave's 64 KT points by dim-2 cell (a hint: NaN -> bin 0)
    cnt[lane] = 0;
    wave_sync();
    unsigned pos[KT];  // rank inside (wave, bin) | bin << 16, then the sorted position
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const T u = (x[k][2] - s.key_start) * s.key_scale;
      int c = u >= (T)1 ? (u < (T)(a.n[2] - 2) ? (int)u : a.n[2] - 2) : 0;
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
    unsigned src[KT / 2];
#pragma unroll
    for (int k = 0; k < KT; ++k) row16[pos[k]] = (unsigned short)((k / PPV) * (64 * PPV) + PPV * lane + (k % PPV));
    wave_sync();
#pragma unroll
    for (int k2 = 0; k2 < KT / 2; ++k2) src[k2] = (unsigned)row16[(2 * k2) * 64 + lane] | ((unsigned)row16[(2 * k2 + 1) * 64 + lane] << 16);
    wave_sync();
    // -- KT rows in sorted order: cubic_brick.h's row (fully overlapped tiles, LDS-DMA gather)
    T res[KT];
    sweep_static_for<0, KT>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      __builtin_amdgcn_sched_barrier(0);
      T xr[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) xr[d] = k < K ? x[k < K ? k : 0][d] : park[(d * KL + (k - K)) * 64 + lane];
      DimT dim[3];
      int loc[3];
      bool ok = true;
      if constexpr (RECT) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const Axis<T> ax = make_axis<T, 3>(a.ax, axis_base, d);
          loc[d] = cubic_rect_locate<T>(ax, xr[d], a.linearize, /*fma_linear=*/false, dim[d]);  // multicubic/rectilinear.rs:366-408
        }
      } else {
        // cell, saturation class and t of the three dimensions: first without the six divide sequences
        // (interpn_device.h::floor_quotient_fast / divide_fast: the same bits where their conditions hold), then, if a
        // lane of the wave is on a grid plane, not finite or far out, once more as the reference writes them
        auto locate_all = [&](auto fast_c) -> bool {
          constexpr bool FAST = decltype(fast_c)::value;
          bool exact = true;
          ok = true;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const T xx = xr[d];
            T floc;
            if constexpr (FAST) {
              exact = floor_quotient_fast(xx - a.start[d], s.rstep[d], &floc) && exact;  // (|floc| < 2^31: representable)
            } else {
              ok &= regular_floc<T>(xx, a.start[d], a.step[d], &floc);   // multicubic/regular.rs:435-438
              ok &= floc != (T)-9223372036854775808.0;                   // `- 1` would overflow isize
            }
            const T nn = (T)a.n[d];
            const int l = clamp_loc<T>(floc - (T)1, a.n[d] - 4);       // regular.rs:440-442
            int sat;
            bool outside;
            if (floc < (T)0) { sat = kSatLow; outside = true; }        // regular.rs:445-466 on floc = iloc + 1
            else if (floc == (T)0) { sat = kSatLow; outside = false; }
            else if (floc > nn - (T)2) { sat = kSatHigh; outside = true; }
            else if (floc == nn - (T)2) { sat = kSatHigh; outside = false; }
            else { sat = kSatNone; outside = false; }
            const T index_one_loc = mul_add<false>(a.step[d], (T)(l + 1), a.start[d]);  // regular.rs:356-360, never fused
            T t;
            if constexpr (FAST) exact = divide_fast(xx - index_one_loc, a.step[d], s.rstep[d], &t) && exact;
            else t = (xx - index_one_loc) / a.step[d];
            dim[d].sat = sat;
            dim[d].linear = (outside && a.linearize) ? 1 : 0;
            dim[d].tt = sat == kSatLow ? -t : (sat == kSatHigh ? t - (T)1 : t);
            loc[d] = l;
          }
          return exact;
        };
        bool exact = false;
        if (s.fastdiv) exact = locate_all(std::true_type{});
        if (__any(!exact)) (void)locate_all(std::false_type{});
        const size_t gi = base + ((src[k / 2] >> (16 * (k & 1))) & 0xFFFFu);
        if (!ok && gi < a.npts) atomicMin(a.first_bad, (unsigned long long)gi);
      }
      // my point's tile (steps 1,1: tile index = cell) as a byte offset; instruction q of a plane's DMA has me
      // fetch piece c of point p (cubic_brick.h::dma_issue_plane)
      constexpr unsigned PP = (unsigned)sizeof(T);
      const unsigned tb = ((unsigned)loc[2] * a.plane_stride[2] + (unsigned)(loc[0] * (int)a.nbj + loc[1]) * 16u) * (unsigned)sizeof(T);
      unsigned toff[PP];
#pragma unroll
      for (int q = 0; q < (int)PP; ++q) {
        const unsigned p = ((unsigned)q * 64u + lane) / PP;
        const unsigned c = ((lane & (PP - 1u)) - cubic_dma_rot<T>(p)) & (PP - 1u);
        toff[q] = (unsigned)__shfl((int)tb, (int)p) + c * 16u;
      }
      unsigned interior = 0;
      if constexpr (!RECT) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
          if (__builtin_amdgcn_ballot_w64(dim[d].sat != kSatNone || dim[d].linear != 0) == 0) interior |= 1u << d;
      }
      res[k] = reduce_planes_dma<T, 3, RECT, FMA>(rsrc, toff, a.plane_stride, lds_wave, lane, dim, interior);
      asm volatile("" : "+v"(res[k]));
    });
    // -- results back into the points' own order through LDS
#pragma unroll
    for (int k = 0; k < KT; ++k) row[(src[k / 2] >> (16 * (k & 1))) & 0xFFFFu] = res[k];
    wave_sync();
#pragma unroll
    for (int kv = 0; kv < KT / PPV; ++kv) {
      const size_t i0 = base + (size_t)kv * (64 * PPV) + PPV * lane;
      const TV v = *reinterpret_cast<const TV*>(&row[kv * (64 * PPV) + PPV * lane]);
      if (full || i0 + PPV - 1 < a.npts) {
        stream_store(reinterpret_cast<TV*>(a.out + i0), v);
      } else {
#pragma unroll
        for (int h = 0; h < PPV; ++h)
          if (i0 + h < a.npts) stream_store(a.out + i0 + h, v[h]);
      }
    }
    wave_sync();
  }
  // -- the period measurement, exactly as in linear_sweep.h
  const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const unsigned my_ticks = (unsigned)(t_end - t_begin);
    atomicAdd(&wg_words[0], my_ticks);
    atomicAdd(&wg_words[1], my_rounds);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (atomicAdd(&wg_words[2], 1u) == THREADS / 64 - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const unsigned wg_ticks = wg_words[0], wg_rounds = wg_words[1];
      if (wg_rounds) {
        const unsigned long long r1 = atomicAdd(&work->ticks, (unsigned long long)wg_ticks);
        const unsigned r2 = atomicAdd(&work->rounds, wg_rounds);
        asm volatile("" ::"v"(r1), "v"(r2));
      }
      const unsigned d = atomicAdd(&work->done[0], 1u);
      if (d == gridDim.x - 1) {
        const unsigned long long ticks = atomicAdd(&work->ticks, 0ull);
        const unsigned rounds = atomicAdd(&work->rounds, 0u);
        if (rounds >= 4 * nwaves) {
          unsigned long long p = ticks * 9 / ((unsigned long long)rounds * 10);
          p = p < 200 ? 200 : (p > 40000 ? 40000 : p);
          atomicExch(&work->period, (unsigned)p);
        }
        for (int xx = 0; xx < 8; ++xx) atomicExch(&work->head[xx][0], 0u);
        atomicExch(&work->ticks, 0ull);
        atomicExch(&work->rounds, 0u);
        atomicExch(&work->done[0], 0u);
      }
    }
  }
}

}  // namespace interpn

// The scaffold of the sweep evaluations (round 5): what linear_sweep.h does around its rows — every wave keeps K x 64 points
// in registers (+ KL x 64 parked in LDS), counting-sorts them by their cell index along the table's slowest dimension through
// its own LDS, walks its rows in that order starting with the row a chip-wide clock names (s_memrealtime: all waves of an XCD
// then gather from the same slab of the table at the same time), puts the results back into the points' own order, and takes
// rounds on demand from eight counters; every launch measures the period for the next (read linear_sweep.h's head for the
// why and the measurements) — as ONE function with the row left to the caller: `row_fn(integral_constant<int, k>, xr, gi)`
// evaluates the wave's row k (xr: the lane's point, gi: its index in the batch) and returns the lane's result.
// Used by cubic_sweep.h (3-D / 2-D multicubic), k_linear2_brick.hip (2-D multilinear) and k_nearest.hip (2-D / 3-D nearest
// neighbour).  linear_sweep.h — the headline kernel, written first — keeps its own copy of these lines.  Round 6 moved it
// onto this function (it compiles, 259 sweep tests bit-identical, the f64 shapes free of scratch with OPAQUE_LANE = 1) and
// measured it on one box against the copy: cfg2 +0.6 % and +1.3 % on two boxes, the f32 shapes 2-9 % slower (24 rows in
// registers spill 3-24 here, 20 + 4 parked do not) — the copy stays (profiles/REJECTED.md); the ABL / STAMPS hooks and the
// CNT_BYTES / OPAQUE_LANE parameters that the move needed stay too.
#pragma once

#include <type_traits>

#include "interpn_device.h"

namespace interpn {

// Work words of the launches through one scratch block (device memory).  Zero before the first
// launch; the last wave of every launch leaves everything but `period` zero again, so launches that
// follow each other on a stream need no reset in between.
struct SweepWork {
  unsigned head[8][32];        // next round of shard x (one 128-byte line each)
  unsigned done[32];           // waves that have finished
  unsigned long long ticks;    // sum over waves of the ticks spent in rounds ...
  unsigned rounds;             // ... and of the rounds they took
  unsigned period;             // ticks per sweep for the next launch (0: the launch's default)
  // the sampling kernel in front of an automatic launch (k_linear_sweep.hip::k_sweep_probe): its counters (left zero)
  // and its verdict for the two gated launches behind it — non-zero: the batch is coherent as it stands, the one-pass
  // kernel takes it and the sweep kernel's waves return at once
  unsigned probe_changes, probe_done, take_brick;
  unsigned pad[25];
};
static_assert(sizeof(SweepWork) == 10 * 128, "one line per counter");


// f(integral_constant<int, I>) for I = FROM .. TO - 1 (a row's index must be a constant: its coordinates live in registers)
template <int FROM, int TO, typename F>
__device__ __forceinline__ void sweep_static_for(F&& f) {
  if constexpr (FROM < TO) {
    f(std::integral_constant<int, FROM>{});
    sweep_static_for<FROM + 1, TO>(f);
  }
}

template <typename T, int N>
struct SweepRounds {
  const T* obs[N];
  T* out;
  size_t npts;
  T absent[N];             // coordinates of the places behind the batch's last point (evaluated like points, never stored)
  T key_start, key_scale;  // the sort key: (x[KEYDIM] - key_start) * key_scale ~ the cell index along the table's slowest dimension (a locality hint)
  int key_cells;           // ... clamped to [0, key_cells]
  int key_shift;           // ... >> key_shift < 64 bins
  unsigned rounds;         // 64 * (K + KL) points each
  unsigned per_shard;      // rounds per shard (8 shards)
  unsigned period;         // > 0: ticks of 10 ns per sweep, overriding the measured one; 1: rows in sorted order (no clock)
  unsigned period_default; // before anything has been measured
  unsigned gated;          // != 0: do nothing if work->take_brick is set (see SweepWork)
  SweepWork* work;
  // measurement builds only (tools/): ABL == 2 makes the coordinates up from the point index on this grid and stores
  // nothing; STAMPS leaves 8 words per wave (start | end | ticks outside the rows | ticks in rows | XCD | last first row |
  // rounds | period)
  T fake_start[N], fake_step[N];
  int fake_n[N];
  unsigned long long* stamps;
};

// LDS per wave: a row buffer (the sort's and the results' exchange; MIN_ROW_BYTES: what the caller's rows need of the same
// bytes in between, e.g. a tile image), the parked rows, the sort's counters.  Behind the waves' regions: 16 bytes of the scaffold.
// CNT_BYTES: the counters' region, larger where the caller's rows use those bytes too (linear_sweep.h: 1 KiB of piece offsets).
template <typename T, int N, int K, int KL, unsigned MIN_ROW_BYTES = 0, unsigned CNT_BYTES = 64u * 4u * 2u>
struct SweepRoundsLds {
  static constexpr unsigned kRowOnly = 64u * K * sizeof(T);
  static constexpr unsigned kRow = kRowOnly > MIN_ROW_BYTES ? kRowOnly : MIN_ROW_BYTES;
  static constexpr unsigned kPark = 64u * KL * (unsigned)N * sizeof(T);
  static constexpr unsigned kCnt = CNT_BYTES;
  static_assert(CNT_BYTES >= 64u * 4u * 2u, "64 counters + 64 first positions");
  static constexpr unsigned kWave = kRow + kPark + kCnt;
  static constexpr unsigned kWorkgroup = 16;
  static_assert(kRow + kPark >= 64u * (K + KL) * sizeof(T), "the result exchange spans the row buffer and the parked rows' bytes");
  static_assert(64u * (K + KL) * 2u <= kRow, "the 16-bit source-index exchange of a round stays inside the row buffer (the parked coordinates sit right behind it)");
};

// Call with the whole workgroup, once, after whatever the kernel stages into LDS behind the waves' regions (the barrier in here
// is the only one).  KEYDIM: the coordinate the points are ordered by.
// OPAQUE_LANE: see the round's first lines (a per-kernel choice: it trades registers held across the persistent loop for a few
// dozen integer instructions per round; which way a shape spills less is found by compiling it, tools/kernel_resources.py)
template <typename T, int N, int K, int KL, int THREADS, int KEYDIM, typename L, int ABL = 0, bool STAMPS = false, int OPAQUE_LANE = 0, typename RowFn>
__device__ __forceinline__ void sweep_rounds(const SweepRounds<T, N>& sr, unsigned char* smem_raw, RowFn&& row_fn) {
  if (sr.gated && __hip_atomic_load(&sr.work->take_brick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // (launch-uniform: no work word is touched)
  constexpr int PPV = 16 / (int)sizeof(T);
  constexpr int KT = K + KL;
  static_assert(KT % PPV == 0 && KT % 2 == 0 && K >= PPV && KT <= 32, "rows per wave and round");
  typedef T TV __attribute__((ext_vector_type(PPV)));
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
  __syncthreads();
  constexpr size_t kChunk = (size_t)64 * KT;
  const unsigned nwaves = gridDim.x * (THREADS / 64);
  SweepWork* const work = sr.work;
  unsigned period = sr.period;
  if (period == 0) {
    period = __hip_atomic_load(&work->period, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (period == 0) period = sr.period_default;
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
  unsigned long long st_mark = t_begin, st_other = 0, st_rows = 0;
  unsigned st_rot = 0;
  unsigned my_rounds = 0;
  __builtin_amdgcn_s_setprio(2);  // wave priority: 2 around the rows, 0 inside them (linear_sweep.h has the measurements)
  unsigned ticket = take(shard);
  while (true) {
    unsigned rr = __builtin_amdgcn_readfirstlane(ticket);
    if (rr >= sr.per_shard || shard * sr.per_shard + rr >= sr.rounds) {
      bool found = false;
      for (unsigned c = 1; c < 8 && !found; ++c) {
        const unsigned sh = (shard + c) & 7u;
        const unsigned seen = __hip_atomic_load(&work->head[sh][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen < sr.per_shard && sh * sr.per_shard + seen < sr.rounds) { found = true; shard = sh; }
      }
      if (!found) break;
      ticket = take(shard);
      continue;
    }
    const unsigned r = shard * sr.per_shard + rr;
    ticket = take(shard);
    ++my_rounds;
    // a copy of the lane id the optimiser cannot see through: what is derived from it inside a round (exchange addresses,
    // source indices: ~40 per-lane constants) is then recomputed per round instead of being hoisted out of the persistent
    // loop into registers the rows need (they spill)
    unsigned ln = lane;
    if constexpr (OPAQUE_LANE == 1) asm volatile("" : "+v"(ln));
    unsigned ln2 = ln;  // (OPAQUE_LANE == 2: only the source indices of the result exchange)
    if constexpr (OPAQUE_LANE == 2) asm volatile("" : "+v"(ln2));
    const size_t base = (size_t)r * kChunk;
    T x[KT][N];
    const bool full = base + kChunk <= sr.npts;
    if (full) {
#pragma unroll
      for (int d = 0; d < N; ++d)
#pragma unroll
        for (int kv = 0; kv < KT / PPV; ++kv) {
          TV v;
          if constexpr (ABL == 2) {
#pragma unroll
            for (int h = 0; h < PPV; ++h) v[h] = ablate_coord<T>(base + (size_t)(kv * 64 + (int)ln) * PPV + h, d, sr.fake_start[d], sr.fake_step[d], sr.fake_n[d]);
          } else {
            v = stream_load(reinterpret_cast<const TV*>(sr.obs[d] + base) + (kv * 64 + (int)ln));
          }
#pragma unroll
          for (int h = 0; h < PPV; ++h) x[PPV * kv + h][d] = v[h];
        }
    } else {
#pragma unroll
      for (int d = 0; d < N; ++d)
#pragma unroll
        for (int kv = 0; kv < KT / PPV; ++kv) {
          const size_t i0 = base + (size_t)kv * (64 * PPV) + PPV * ln;
          TV v;
#pragma unroll
          for (int h = 0; h < PPV; ++h) v[h] = sr.absent[d];
          if (i0 + PPV - 1 < sr.npts) {
            v = stream_load(reinterpret_cast<const TV*>(sr.obs[d] + i0));
          } else {
#pragma unroll
            for (int h = 0; h < PPV; ++h)
              if (i0 + h < sr.npts) v[h] = stream_load(sr.obs[d] + i0 + h);
          }
#pragma unroll
          for (int h = 0; h < PPV; ++h) x[PPV * kv + h][d] = v[h];
        }
    }
    cnt[ln] = 0;
    wave_sync();
    unsigned pos[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const T u = (x[k][KEYDIM] - sr.key_start) * sr.key_scale;
      int c = u >= (T)1 ? (u < (T)sr.key_cells ? (int)u : sr.key_cells) : 0;
      const unsigned bin = (unsigned)(c >> sr.key_shift);
      pos[k] = atomicAdd(&cnt[bin], 1u) | (bin << 16);
    }
    wave_sync();
    {
      const unsigned mine_cnt = cnt[ln];
      unsigned incl = mine_cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, off);
        if (ln >= (unsigned)off) incl += up;
      }
      cnt[64 + ln] = incl - mine_cnt;
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
    for (int d = 0; d < N; ++d) {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        unsigned at = pos[k];
        if constexpr (KL > 0) at += pos[k] >= 64u * K ? kParkSkip + (unsigned)d * (64u * KL) : 0u;  // (straight to where the parked rows wait)
        row[at] = x[k][d];
      }
      wave_sync();
#pragma unroll
      for (int k = 0; k < K; ++k) x[k][d] = row[k * 64 + ln];
      wave_sync();
    }
    unsigned src[KT / 2];
#pragma unroll
    for (int k = 0; k < KT; ++k) row16[pos[k]] = (unsigned short)((k / PPV) * (64 * PPV) + PPV * ln2 + (k % PPV));
    wave_sync();
#pragma unroll
    for (int k2 = 0; k2 < KT / 2; ++k2) src[k2] = (unsigned)row16[(2 * k2) * 64 + ln] | ((unsigned)row16[(2 * k2 + 1) * 64 + ln] << 16);
    wave_sync();
    if constexpr (STAMPS) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); st_other += now - st_mark; st_mark = now; st_rot = rot; }
    __builtin_amdgcn_s_setprio(0);
    T res[KT];
    sweep_static_for<0, KT>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      __builtin_amdgcn_sched_barrier(0);
      T xr[N];
#pragma unroll
      for (int d = 0; d < N; ++d) xr[d] = k < K ? x[k < K ? k : 0][d] : park[(d * KL + (k - K)) * 64 + ln];
      const size_t gi = base + ((src[k / 2] >> (16 * (k & 1))) & 0xFFFFu);  // the point's index in the batch (>= npts: a place behind its end)
      res[k] = row_fn(kc, xr, gi);
      asm volatile("" : "+v"(res[k]));
    });
    if constexpr (STAMPS) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); st_rows += now - st_mark; st_mark = now; }
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int k = 0; k < KT; ++k) row[(src[k / 2] >> (16 * (k & 1))) & 0xFFFFu] = res[k];
    wave_sync();
#pragma unroll
    for (int kv = 0; kv < KT / PPV; ++kv) {
      const size_t i0 = base + (size_t)kv * (64 * PPV) + PPV * ln;
      const TV v = *reinterpret_cast<const TV*>(&row[kv * (64 * PPV) + PPV * ln]);
      if constexpr (ABL == 2) {
        if (v[0] == (T)123.456) stream_store(reinterpret_cast<TV*>(sr.out + i0), v);  // (never)
      } else if (full || i0 + PPV - 1 < sr.npts) {
        stream_store(reinterpret_cast<TV*>(sr.out + i0), v);
      } else {
#pragma unroll
        for (int h = 0; h < PPV; ++h)
          if (i0 + h < sr.npts) stream_store(sr.out + i0 + h, v[h]);
      }
    }
    wave_sync();
  }
  const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {  // the period measurement, exactly as in linear_sweep.h
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
          p = p < 200 ? 200 : (p > 40000 ? 40000 : p);  // 2 .. 400 us
          atomicExch(&work->period, (unsigned)p);
        }
        for (int xx = 0; xx < 8; ++xx) atomicExch(&work->head[xx][0], 0u);
        atomicExch(&work->ticks, 0ull);
        atomicExch(&work->rounds, 0u);
        atomicExch(&work->done[0], 0u);
      }
    }
    if constexpr (STAMPS) {
      if (sr.stamps) {
        unsigned long long* o = sr.stamps + (size_t)(blockIdx.x * (THREADS / 64) + wave) * 8;
        o[0] = t_begin; o[1] = t_end; o[2] = st_other; o[3] = st_rows;
        o[4] = xcc & 0xFu; o[5] = st_rot / 64u; o[6] = my_rounds; o[7] = period;
      }
    }
  }
}

}  // namespace interpn

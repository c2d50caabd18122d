// dev_tile.h -- the wavefront tile in HBM: storage modes, stream hints, TileIO (one env = 4 x 16 B in, the same out).
// Device code of copterstep_kernels.hip (included there, inside its floating-point-contraction pragma);
// not a stand-alone header.
#pragma once

namespace cs {
namespace {

// Diagnostic build only (make stamps): per-wavefront shader-clock stamps at phase
// boundaries, written to a side buffer that nothing else reads.  Never defined in the
// product library.
#ifdef CS_STAMPS
#define CS_STAMP(slot)                                                              \
  do {                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                              \
    unsigned long long t_;                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                              \
    if (lane == 0 && s.stamps) s.stamps[(size_t)tile_index * 8 + (slot)] = t_;      \
  } while (0)
#else
#define CS_STAMP(slot) ((void)0)
#endif

// Diagnostic build only (make kstamps): shader-clock stamps at the phase boundaries of ONE loop iteration of a K-step
// kernel (and the one after it), through a pointer that is null in every other iteration.  The stamp is one asm
// statement between two scheduling barriers (MI355X guide: s_memtime + its own lgkmcnt(0)); it pins the phases in
// program order -- the un-instrumented kernel lets the scheduler interleave them -- so read the shares, and the
// cost of the stamps themselves from the back-to-back pair (slots 14 / 15).
#ifdef CS_KSTAMPS
#define CS_KSTAMP(kst, slot)                                                        \
  do {                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                              \
    unsigned long long t_;                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                              \
    if ((kst) != nullptr && threadIdx.x == 0) (kst)[slot] = t_;                     \
  } while (0)
// the chip-wide 100 MHz clock into a slot (with the s_memtime stamp of the same place: the clock the kernel runs at)
#define CS_KSTAMP_REALTIME(kst, slot)                                               \
  do {                                                                              \
    unsigned long long t_;                                                          \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");   \
    if ((kst) != nullptr && threadIdx.x == 0) (kst)[slot] = t_;                     \
  } while (0)
#else
#define CS_KSTAMP(kst, slot) ((void)0)
#define CS_KSTAMP_REALTIME(kst, slot) ((void)0)
#endif

#ifdef CS_SPAN
#define CS_SPAN_BEGIN() const unsigned long long span_t0_ = __builtin_amdgcn_s_memrealtime()
#define CS_SPAN_END()                                                                                         \
  do {                                                                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the wavefront's stores have been acknowledged */      \
    if (lane == 0 && s.span) {                                                                                \
      unsigned long long* sp_ = s.span + ((size_t)s.span_slot * s.ntiles + tile_index) * 2;                   \
      sp_[0] = span_t0_;                                                                                      \
      sp_[1] = __builtin_amdgcn_s_memrealtime();                                                              \
    }                                                                                                         \
  } while (0)
#else
#define CS_SPAN_BEGIN() ((void)0)
#define CS_SPAN_END() ((void)0)
#endif

constexpr int kBlock = 64;  // one wavefront = one tile = one workgroup (measured best at 65 536 envs: tools/ab.sh)
constexpr int kWave = 64;

template <int MODE>
struct ModeOf {
  using T = float;
  using W = uint32_t;  // a state word as raw bits
  static constexpr Layout L = make_layout(MODE);
};
template <>
struct ModeOf<CS_STATE_F64> {
  using T = double;
  using W = unsigned long long;
  static constexpr Layout L = make_layout(CS_STATE_F64);
};

__device__ __forceinline__ float as_word(uint32_t w) { return __uint_as_float(w); }
__device__ __forceinline__ double as_word(unsigned long long w) { return __longlong_as_double((long long)w); }
__device__ __forceinline__ uint32_t as_bits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ unsigned long long as_bits(double v) { return (unsigned long long)__double_as_longlong(v); }

// caller-owned arrays: uniform base + 32-bit byte offset (global saddr + voffset addressing)
template <class U, class P>
__device__ __forceinline__ U* at32(P* base, uint32_t byte_off) {
  return reinterpret_cast<U*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>(base)) + byte_off);
}

// Streaming accesses (non-temporal hint).  The per-step outputs (observation rows, reward,
// flags) pass through once: stored as streams they do not displace the env state in the caches
// (the per-XCD L2s are written back and invalidated at every kernel boundary; what carries the
// state from one launch to the next is the 256 MiB Infinity Cache) -- measured -5 % time from
// 131 072 to 1 M envs.  Action rows are loaded as streams only for small batches
// (Tuning::nt_action_max_envs): -3 % at 65 536 envs, whether the actions come from a long resident
// ring or were just written by a kernel (scripts/ab_action_source.sh); for larger batches a
// non-temporal load is slower than a plain one (+2..7 %).  tools/ab.sh.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CS_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
constexpr uint32_t kNtActionMaxEnvs = 98304;
// From this batch size the env state no longer fits the 256 MiB Infinity Cache;
// streaming it (non-temporal loads and stores) measured -17 % time at 4 M envs and -14 % at 16 M
// under reset churn, but +7..25 % at 1 M envs and below, where the caches do hold it.
constexpr uint32_t kNtStateMinEnvs = 3670016;  // 3.5 M (3 M envs still measured 5..10 % better un-streamed)
// K-step kernels: up to this many envs (one wavefront per SIMD on 256 CUs) the observation rows are stored
// per lane instead of through the LDS transpose: -8..12 % per step at 65 536 envs, +13..55 % from 131 072 up
constexpr uint32_t kDirectRowsMaxEnvs = 65536;
template <bool STREAM, class V>
__device__ __forceinline__ V load_maybe_stream(const V* p) {
  if constexpr (STREAM) {
    return __builtin_nontemporal_load(p);
  } else {
    return *p;
  }
}

template <class T>
struct alignas(4 * sizeof(T)) Vec4 {
  T v[4];
};

// Per-lane view of one tile.  The tile base is wave-uniform (64-bit, scalar registers), the lane
// offset is 32-bit and biased by kBias so that every field offset fits the signed 13-bit immediate of
// global_load/store (float32 modes), and a whole 4-word group moves as one 16-byte-per-lane
// instruction.
constexpr int kBias = 4096;

// STREAM: the state groups are accessed with the non-temporal hint (batches whose state exceeds the
// 256 MiB Infinity Cache: see launch_step).
template <int MODE, bool STREAM = false>
struct TileIO {
  using T = typename ModeOf<MODE>::T;
  using W = typename ModeOf<MODE>::W;
  using Group = Vec4<W>;
  static constexpr Layout L = ModeOf<MODE>::L;
  char* bg;  // lane stride 4*word  (T1, T2, R1, R2, FE groups)
  char* b4;  // lane stride 4       (RET row)

  __device__ __forceinline__ TileIO(const DevState& s, uint32_t tile, uint32_t lane) {
    char* tb = s.tiles + (size_t)tile * L.tile_bytes;  // wave-uniform: scalar arithmetic, 64-bit
    bg = tb + (uint32_t)(kBias + lane * (4u * L.word));
    b4 = tb + (uint32_t)(kBias + lane * 4u);
  }
  template <class U>
  static __device__ __forceinline__ U ld(const char* p, uint32_t off) {
    if constexpr (STREAM && sizeof(U) == 16) {
      const f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + ((int)off - kBias)));
      U u;
      __builtin_memcpy(&u, &r, 16);
      return u;
    } else {
      return *reinterpret_cast<const U*>(p + ((int)off - kBias));
    }
  }
  template <class U>
  static __device__ __forceinline__ void st(char* p, uint32_t off, const U& v) {
    if constexpr (STREAM && sizeof(U) == 16) {
      f32x4 r;
      __builtin_memcpy(&r, &v, 16);
      __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(p + ((int)off - kBias)));
    } else {
      *reinterpret_cast<U*>(p + ((int)off - kBias)) = v;
    }
  }
  __device__ __forceinline__ Group load_group(int j) const { return ld<Group>(bg, L.grp[j]); }
  __device__ __forceinline__ void store_group(int j, const Group& g) const { st(bg, L.grp[j], g); }
  __device__ __forceinline__ float load_ret() const { return ld<float>(b4, L.ret); }
  __device__ __forceinline__ void store_ret(float v) const { st(b4, L.ret, v); }
  // EPH row: the episode counter's bits above the meta word's (plain accesses: once per 2^ep_bits episodes of an env)
  __device__ __forceinline__ uint32_t load_eph() const {
    return *reinterpret_cast<const uint32_t*>(b4 + ((int)L.eph - kBias));
  }
  __device__ __forceinline__ void store_eph(uint32_t hi) const {
    *reinterpret_cast<uint32_t*>(b4 + ((int)L.eph - kBias)) = hi;
  }
  // FE group: the EXPLICIT pending force [N] in its first three words (plain accesses: rare); the fourth
  // word is the Dynamics tick counter (cs_config.track_time), so a force is stored as three words
  __device__ __forceinline__ Vec4<T> load_fe() const {
    return *reinterpret_cast<const Vec4<T>*>(bg + ((int)L.fe - kBias));
  }
  __device__ __forceinline__ void store_fe(const Vec4<T>& v) const {
    T* p = reinterpret_cast<T*>(bg + ((int)L.fe - kBias));
    p[0] = v.v[0];
    p[1] = v.v[1];
    p[2] = v.v[2];
  }
  // Dynamics._ticks (dynamics/__init__.py:98, :197): setMotors calls of this episode that did not freeze on
  // ground contact.  Kept only under cs_config.track_time (full-featured kernels).
  __device__ __forceinline__ uint32_t load_ticks() const {
    return *reinterpret_cast<const uint32_t*>(bg + ((int)(L.fe + 3u * L.word) - kBias));
  }
  __device__ __forceinline__ void store_ticks(uint32_t t) const {
    *reinterpret_cast<uint32_t*>(bg + ((int)(L.fe + 3u * L.word) - kBias)) = t;
  }

  // the integer words of the T2 group (gT, meta) and of the R2 group (gR), and R2's fourth word: prev_shaping
  static __device__ __forceinline__ uint32_t int_lo(const Group& g) {  // gT or gR
    return (uint32_t)g.v[2];
  }
  static __device__ __forceinline__ uint32_t int_hi(const Group& g) {  // meta (T2 only)
    if constexpr (sizeof(W) == 4) {
      return g.v[3];
    } else {
      return (uint32_t)(g.v[2] >> 32);
    }
  }
  static __device__ __forceinline__ void set_t2(Group& g, uint32_t gT, uint32_t meta) {
    if constexpr (sizeof(W) == 4) {
      g.v[2] = gT;
      g.v[3] = meta;
    } else {
      g.v[2] = (W)gT | ((W)meta << 32);
      g.v[3] = 0;
    }
  }
  static __device__ __forceinline__ T prev_of(const Group& r2) { return as_word(r2.v[3]); }
  static __device__ __forceinline__ void set_r2(Group& g, uint32_t gR, T prev) {
    g.v[2] = (W)gR;
    g.v[3] = as_bits(prev);
  }
};

}  // namespace
}  // namespace cs

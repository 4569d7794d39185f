// dev_load.h — typed, vectorised column loads/stores shared by the kernels.
// Operands are loaded with the widest natural vector load chosen by a
// wave-uniform switch and widened losslessly to 64 bits.
#pragma once
#include "dev_common.h"

enum { CLS_SIGNED = 0, CLS_UNSIGNED = 1, CLS_FLOAT = 2 };

__host__ __device__ inline int type_class(int t) {
  switch (t) {
    case DBHIP_T_I8: case DBHIP_T_I16: case DBHIP_T_I32: case DBHIP_T_I64:
    case DBHIP_T_DATE: case DBHIP_T_TIMESTAMP: case DBHIP_T_DEC64:
      return CLS_SIGNED;
    case DBHIP_T_U8: case DBHIP_T_U16: case DBHIP_T_U32: case DBHIP_T_U64:
      return CLS_UNSIGNED;
    case DBHIP_T_F32: case DBHIP_T_F64:
      return CLS_FLOAT;
    default:
      return -1;
  }
}

__host__ __device__ inline int type_bits(int t) {
  switch (t) {
    case DBHIP_T_I8: case DBHIP_T_U8: return 8;
    case DBHIP_T_I16: case DBHIP_T_U16: return 16;
    case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE: return 32;
    default: return 64;
  }
}

template <typename T, int N>
struct alignas(sizeof(T) * N) VecT {
  T v[N];
};

// Load 4 consecutive elements starting at i0 (i0 % 4 == 0), widened to 64 bits:
// signed -> sign-extended, unsigned -> zero-extended, f32/f64 -> f64 bit pattern.
template <typename T>
__device__ __forceinline__ void load4_t(const void* p, bool scalar, int64_t i0, int64_t n,
                                        uint64_t out[4]) {
  const T* q = (const T*)p;
  T tmp[4];
  if (scalar) {
    T s = q[0];
    tmp[0] = tmp[1] = tmp[2] = tmp[3] = s;
  } else if (i0 + 4 <= n) {
    VecT<T, 4> v = *(const VecT<T, 4>*)(q + i0);
    tmp[0] = v.v[0]; tmp[1] = v.v[1]; tmp[2] = v.v[2]; tmp[3] = v.v[3];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) tmp[k] = (i0 + k < n) ? q[i0 + k] : T(0);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if constexpr (sizeof(T) == 4 && !__is_integral(T)) {
      out[k] = (uint64_t)__double_as_longlong((double)tmp[k]);
    } else if constexpr (!__is_integral(T)) {
      out[k] = (uint64_t)__double_as_longlong((double)tmp[k]);
    } else if constexpr (((T)-1) < (T)0) {
      out[k] = (uint64_t)(int64_t)tmp[k];
    } else {
      out[k] = (uint64_t)tmp[k];
    }
  }
}

__device__ __forceinline__ void load4_wide(const void* p, int type, bool scalar, int64_t i0,
                                           int64_t n, uint64_t out[4]) {
  switch (type) {
    case DBHIP_T_I8: load4_t<int8_t>(p, scalar, i0, n, out); break;
    case DBHIP_T_I16: load4_t<int16_t>(p, scalar, i0, n, out); break;
    case DBHIP_T_I32: case DBHIP_T_DATE: load4_t<int32_t>(p, scalar, i0, n, out); break;
    case DBHIP_T_I64: case DBHIP_T_TIMESTAMP: case DBHIP_T_DEC64:
      load4_t<int64_t>(p, scalar, i0, n, out); break;
    case DBHIP_T_U8: load4_t<uint8_t>(p, scalar, i0, n, out); break;
    case DBHIP_T_U16: load4_t<uint16_t>(p, scalar, i0, n, out); break;
    case DBHIP_T_U32: load4_t<uint32_t>(p, scalar, i0, n, out); break;
    case DBHIP_T_U64: load4_t<uint64_t>(p, scalar, i0, n, out); break;
    case DBHIP_T_F32: load4_t<float>(p, scalar, i0, n, out); break;
    default: load4_t<double>(p, scalar, i0, n, out); break;
  }
}

template <typename T>
__device__ __forceinline__ void store4_t(void* p, int64_t i0, int64_t n, const T vals[4]) {
  T* q = (T*)p;
  if (i0 + 4 <= n) {
    VecT<T, 4> v;
    v.v[0] = vals[0]; v.v[1] = vals[1]; v.v[2] = vals[2]; v.v[3] = vals[3];
    *(VecT<T, 4>*)(q + i0) = v;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + k < n) q[i0 + k] = vals[k];
  }
}

// Store 4 widened results truncated to the output type (integers wrap; a float
// output carries f64 bits).
__device__ __forceinline__ void store4_wide(void* p, int type, int64_t i0, int64_t n,
                                            const uint64_t r[4]) {
  switch (type) {
    case DBHIP_T_I8: case DBHIP_T_U8: {
      uint8_t v[4] = {(uint8_t)r[0], (uint8_t)r[1], (uint8_t)r[2], (uint8_t)r[3]};
      store4_t<uint8_t>(p, i0, n, v);
    } break;
    case DBHIP_T_I16: case DBHIP_T_U16: {
      uint16_t v[4] = {(uint16_t)r[0], (uint16_t)r[1], (uint16_t)r[2], (uint16_t)r[3]};
      store4_t<uint16_t>(p, i0, n, v);
    } break;
    case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_DATE: {
      uint32_t v[4] = {(uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]};
      store4_t<uint32_t>(p, i0, n, v);
    } break;
    case DBHIP_T_F32: {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (float)__longlong_as_double((long long)r[k]);
      store4_t<float>(p, i0, n, v);
    } break;
    default: {
      uint64_t v[4] = {r[0], r[1], r[2], r[3]};
      store4_t<uint64_t>(p, i0, n, v);
    } break;
  }
}


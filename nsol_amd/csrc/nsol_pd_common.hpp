// Helpers shared by the single-pass primal-dual kernels (nsol_pd.hip,
// nsol_pd2.hip).
#pragma once

#include "nsol_common.hpp"

namespace nsol {

template <typename T, int V>
struct Pack {
  typedef T type __attribute__((ext_vector_type(V)));
};

template <typename T, int V>
__device__ __forceinline__ void ldv(const T *p, T (&v)[V]) {
  if constexpr (V == 1) {
    v[0] = *p;
  } else {
    typedef typename Pack<T, V>::type P;
    const P t = *reinterpret_cast<const P *>(p);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = t[k];
  }
}

template <typename T, int V>
__device__ __forceinline__ void stv(T *p, const T (&v)[V]) {
  if constexpr (V == 1) {
    *p = v[0];
  } else {
    typedef typename Pack<T, V>::type P;
    P t;
#pragma unroll
    for (int k = 0; k < V; ++k) t[k] = v[k];
    *reinterpret_cast<P *>(p) = t;
  }
}

// The same accesses for rows whose length is not a multiple of V elements (or
// whose base is not 16-byte aligned): the vector starts at an element-aligned
// address (legal for global memory, ~90 % of the aligned rate) and the row's last
// vector holds nval < V valid elements, moved one by one; what lies behind the row
// reads as zero and is never written.
template <typename T, int V>
__device__ __forceinline__ void ldv_rag(const T *p, T (&v)[V], int nval) {
  typedef T P __attribute__((ext_vector_type(V), aligned(sizeof(T))));
  if (nval >= V) {
    const P t = *reinterpret_cast<const P *>(p);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = t[k];
  } else {
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = k < nval ? p[k] : T(0);
  }
}

template <typename T, int V>
__device__ __forceinline__ void stv_rag(T *p, const T (&v)[V], int nval) {
  typedef T P __attribute__((ext_vector_type(V), aligned(sizeof(T))));
  if (nval >= V) {
    P t;
#pragma unroll
    for (int k = 0; k < V; ++k) t[k] = v[k];
    *reinterpret_cast<P *>(p) = t;
  } else {
#pragma unroll
    for (int k = 0; k < V; ++k)
      if (k < nval) p[k] = v[k];
  }
}

template <typename T, int V>
__device__ __forceinline__ void zero(T (&v)[V]) {
#pragma unroll
  for (int k = 0; k < V; ++k) v[k] = T(0);
}

template <typename T>
struct PdScalars {
  T sigma, hden, tau, tl, one_plus_tl, theta;
  int huber, l1, has_p;
};

// p_new = clamp((p_old + sigma * (hi*w + lo*(-w))) / hden)
template <typename T>
__device__ __forceinline__ T dual_update(T p_old, T hi, T lo, T w,
                                         const PdScalars<T> &S) {
  T q = p_old + S.sigma * (hi * w + lo * (-w));
  if (S.huber) {
    pin(q);            // a real (uniform) branch, not a speculated division
    q = huber_div(q, S.hden);
  }
  return dual_clamp(q);
}

// compile-time flags and an explicit step size (sigma may be masked to zero
// for voxels outside the volume, which makes the new dual exactly zero)
template <bool HUBER, typename T>
__device__ __forceinline__ T dual_update_s(T p_old, T hi, T lo, T w, T sigma,
                                           T hden) {
  T q = p_old + sigma * (hi * w + lo * (-w));
  if constexpr (HUBER) q = huber_div(q, hden);
  return dual_clamp(q);
}

// UNIT: all inverse spacings are exactly 1.  x*1 and x*(-1) are exact, so
// hi*1 + lo*(-1) == hi - lo and p*(-1) + pl*1 == pl - p bit for bit: the
// multiplications can be dropped without changing a single result.
template <bool HUBER, bool UNIT, typename T>
__device__ __forceinline__ T dual_update_u(T p_old, T hi, T lo, T w, T sigma, T hden) {
  const T g = UNIT ? hi - lo : hi * w + lo * (-w);
  T q = p_old + sigma * g;
  if constexpr (HUBER) q = huber_div(q, hden);
  return dual_clamp(q);
}
// one axis of K^T p: p*(-w) + p_prev*w
template <bool UNIT, typename T>
__device__ __forceinline__ T adj_term(T p, T p_prev, T w) {
  return UNIT ? p_prev - p : p * (-w) + p_prev * w;
}

template <bool L1, typename T>
__device__ __forceinline__ T prox_data_s(T u, T bt, T tl, T one_plus_tl) {
  if constexpr (L1) return prox_ell1(u, bt, tl);
  else return prox_ell2(u, bt, tl, one_plus_tl);
}

}  // namespace nsol

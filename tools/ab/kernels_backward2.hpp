// kernels_backward2.hpp — A/B ALTERNATIVE, PROFILING BUILDS ONLY (tools/build_profiling_lib.sh, -DHS_PROFILE_HOOKS=1; included by
// hyperslam_amd/csrc/kernels.hpp behind that switch): the backward sweeps that take one block row per step (k_band_backward2,
// HS_DEBUG_FLAGS 268435456 / 8192) and k_step_outputs, the measured alternative of k_band_backward_sb. Not part of the product library.
#pragma once
namespace hs {

__global__ void __launch_bounds__(kCholThreads) k_band_backward2(Tables T, BackJob j0, BackJob j1, int m_mid) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const BackJob J = blockIdx.x == 0 ? j0 : j1;
  const int tid = threadIdx.x;
  constexpr int nthr = kCholThreads;
  const int bw = T.bw, ncb = 6 * bw, np = T.np;
  const bool cprof = prof_enabled(T.debug_flags, 16) && tid == 0;  // coarse phases -> xpart[8 (230 + 10 block) + ..] (tools/chol_phase_timing.py)
  long long* clog = reinterpret_cast<long long*>(T.xpart) + 8 * (230 + 10 * blockIdx.x);
  if (cprof) clog[0] = wall_clock64();
  const int n_own = 6 * J.n_rows, n_all = 6 * (J.n_rows + J.given);
  double* xs = smem;          // n_all : pending rows (own) / given solution
  double* xout = smem + n_all;  // n_own : solution of the own rows (flushed to T.xsol at the end / when the middle is complete)
  __shared__ double Wl[2][24];
  const int n_above = 6 * (bw - 1);
  // The given block rows have no dependencies among themselves: their whole contribution to the pending rows is one
  // (n_above x n_above) matrix-vector product. The matrix block G = U(own rows n_own - n_above .., given columns) does not depend on
  // the other sweep, so it is brought into LDS (transposed: G[c][r], odd leading dimension) BEFORE waiting for the flag; once the
  // middle solution is there, one pass replaces `given` sequential steps of the sweep.
  const bool merged = J.given > 0 && 6 * J.given == n_above && n_own >= n_above;
  const int ldg = n_above | 1;
  double* G = smem + 2 * np;
  for (int rho = tid; rho < n_own; rho += nthr) xs[rho] = J.ybuf[rho];
  if (merged) {
    const int n_g = n_above * n_above;
    for (int e0 = tid; e0 < n_g; e0 += 8 * nthr) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * nthr, r = e / n_above, c = e - r * n_above;
        const int rho = n_own - n_above + r, off = n_own + c - 6 * (rho / 6);  // band offset of column n_own + c in row rho
        v[u] = (e < n_g && off < ncb) ? J.Ub[size_t(rho) * ncb + off] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * nthr, r = e / n_above, c = e - r * n_above;
        if (e < n_g) G[c * ldg + r] = v[u];
      }
    }
  }
  auto load_u = [&](int j, double* u) {
    const int rho = 6 * j - 1 - tid;
    const bool ok = j >= 0 && tid < n_above && rho >= 0 && rho < n_own;
    const double* src = J.Ub + (ok ? size_t(rho) * ncb + (6 * j - 6 * (rho / 6)) : 0);
#pragma unroll
    for (int a = 0; a < 6; ++a) u[a] = ok ? src[a] : 0.0;
  };
  auto load_w = [&](int j) -> double { return (j >= 0 && j < J.n_rows && tid < 21) ? J.Ubk[size_t(j) * 24 + tid] : 0.0; };
  const int jtop = merged ? J.n_rows - 1 : J.n_rows + J.given - 1;
  double u0[6], u1[6], u2[6], u3[6], w0, w1, w2, w3;
  load_u(jtop, u0), load_u(jtop - 1, u1), load_u(jtop - 2, u2);
  w0 = load_w(jtop), w1 = load_w(jtop - 1), w2 = load_w(jtop - 2);
  if (cprof) clog[1] = wall_clock64();  // operands staged
  if (J.given) {  // wait for the middle solution
    wait_for_partner(T);
    if (cprof) clog[2] = wall_clock64();  // middle solution arrived
    for (int rho = n_own + tid; rho < n_all; rho += nthr) xs[rho] = T.xsol[J.reversed ? np - 1 - rho : rho];
  }
  __syncthreads();
  if (merged) {
    if (tid < n_above) {
      double acc = 0.0;
      for (int c = 0; c < n_above; ++c) acc = fma(G[c * ldg + tid], xs[n_own + c], acc);
      xs[n_own - n_above + tid] -= acc;
    }
    __syncthreads();
  }
  // one block row of the sweep; `own` is a compile-time tag so that the hot loops below carry no extra control flow
  auto step = [&](int j, auto own_tag) {
    constexpr bool own = decltype(own_tag)::value;
    if (own && tid < 21) Wl[j & 1][tid] = w0;
    load_u(j - 3, u3), w3 = load_w(j - 3);
    lds_barrier();
    double y[6], x[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) y[a] = xs[6 * j + a];
    if (own) {
      const double* W = Wl[j & 1];
      int pidx = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double v = 0.0;
#pragma unroll
        for (int c = a; c < 6; ++c) v = fma(W[pidx++], y[c], v);
        x[a] = v;
      }
      if (tid < 6) xout[6 * j + tid] = x[tid];
    } else {
#pragma unroll
      for (int a = 0; a < 6; ++a) x[a] = y[a];  // given by the other sweep
    }
    const int rho_p = 6 * j - 1 - tid;
    if (tid < n_above && rho_p >= 0 && rho_p < n_own) {
      double sacc = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) sacc = fma(u0[a], x[a], sacc);
      xs[rho_p] -= sacc;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) u0[a] = u1[a], u1[a] = u2[a], u2[a] = u3[a];
    w0 = w1, w1 = w2, w2 = w3;
  };
  if (cprof) clog[3] = wall_clock64();  // sweep starts
  for (int j = jtop; j >= J.n_rows; --j) step(j, std::false_type{});  // (not merged: given rows one by one)
  const int j_pub = (blockIdx.x == 0 && m_mid >= 0) ? m_mid : 0;  // block 0 publishes the middle solution after block row m_mid
  for (int j = J.n_rows - 1; j >= j_pub; --j) step(j, std::true_type{});
  if (blockIdx.x == 0 && m_mid >= 0) {
    lds_barrier();
    if (cprof) clog[4] = wall_clock64();  // middle rows solved
    for (int rho = 6 * m_mid + tid; rho < n_own; rho += nthr) T.xsol[rho] = xout[rho];
    __threadfence();
    lds_barrier();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (cprof) clog[5] = wall_clock64();  // middle solution published
    for (int j = m_mid - 1; j >= 0; --j) step(j, std::true_type{});
  }
  __syncthreads();
  if (cprof) clog[6] = wall_clock64();  // sweep done
  const int flush_to = (blockIdx.x == 0 && m_mid >= 0) ? 6 * m_mid : n_own;  // (the middle rows of block 0 are already out)
  for (int rho = tid; rho < flush_to; rho += nthr) T.xsol[J.reversed ? np - 1 - rho : rho] = xout[rho];
  if (gridDim.x == 1) return;  // (A/B runs on the whole system: k_step_outputs follows)
  // the block that finishes last turns the solution into the step outputs (saves a launch); join_flag[1] advances by two per launch
  __shared__ int is_last;
  __threadfence();
  __syncthreads();
  if (tid == 0) is_last = (atomicAdd(T.join_flag + 1, 1u) & 1u) == 1u;
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  __shared__ double red[kCholThreads / 64];
  double gd = 0.0, dd = 0.0;
  // (four rows per lane in flight: the other block's half of the solution comes through HBM, a plain loop paid that latency per pass:
  //  5.6 us for 768 rows)
  for (int rho0 = tid; rho0 < np; rho0 += 4 * nthr) {
    double xv[4], sc[4], gf[4], d2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rho = rho0 + u * nthr, rr = rho < np ? rho : 0;
      xv[u] = __builtin_nontemporal_load(T.xsol + rr), sc[u] = T.scale_p[rr], gf[u] = T.g_full[rr], d2[u] = T.D2p[rr];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rho = rho0 + u * nthr;
      if (rho < np) {
        const double step = -xv[u];
        T.step_p[rho] = step;
        T.delta_p[rho] = sc[u] * step;
        gd = fma(gf[u], step, gd);
        dd = fma(d2[u] * step, step, dd);
      }
    }
  }
  gd = block_sum(gd, red);
  dd = block_sum(dd, red);
  if (tid == 0) {
    st->g_dot_step_pose = gd;
    st->d2_step2_pose = dd;
  }
  if (cprof) clog[7] = wall_clock64();  // step outputs written (the block that finished last)
}


/// step = -x, delta = scale o step and the pose-side reductions of the model cost change, from T.xsol (two-ended path).
__global__ void __launch_bounds__(kBlock) k_step_outputs(Tables T) {
  __shared__ double red[kBlock / 64];
  DevState* st = T.st;
  if (st->done) return;
  double gd = 0.0, dd = 0.0;
  for (int rho = threadIdx.x; rho < T.np; rho += kBlock) {
    const double step = -T.xsol[rho];
    T.step_p[rho] = step;
    T.delta_p[rho] = T.scale_p[rho] * step;
    gd = fma(T.g_full[rho], step, gd);
    dd = fma(T.D2p[rho] * step, step, dd);
  }
  gd = block_sum(gd, red);
  dd = block_sum(dd, red);
  if (threadIdx.x == 0) {
    st->g_dot_step_pose = gd;
    st->d2_step2_pose = dd;
  }
}


}  // namespace hs

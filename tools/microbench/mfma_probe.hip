// Probe: is the f64 matrix instruction (v_mfma_f64_16x16x4_f64) worth using for the trailing update of the banded Cholesky?
//   1. lane <-> element layout check of A / B / C (A = I with an asymmetric B, then a random product against a host reference)
//   2. issue interval (independent accumulators) and dependent-accumulator latency on 1 .. 8 waves of one workgroup
//   3. a realistic step of a ring-tiled trailing window: W x W window (W = 96 / 144 / 256) kept in 16x16 accumulator tiles spread
//      over NW waves, rank-K update (K = 8: two MFMAs per tile) with the operands read from LDS (one ds_read_b64 per lane and
//      operand), one LDS-only barrier per step  ->  clk per step, to be compared with p2_probe (2200 clk per 6x6 VALU tile update)
//   4. the panel chain of an 8x8 pivot block in one wave: redundant register Cholesky (rsq + one third-order step) + two column
//      solves per lane, in the plain order and with the column solves interleaved into the factorisation
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- 1. layout: D = A (16 x 4) * B (4 x 16) + C -------------------------------------------------------------------------------
__global__ void k_layout(const double* A /* 16 x 4 row-major */, const double* B /* 4 x 16 row-major */, double* D /* 16 x 16 */) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];   // A[i = l & 15][k = l >> 4]
  const double b = B[(l >> 4) * 16 + (l & 15)];  // B[k = l >> 4][j = l & 15]
  double4_t c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];  // D[row = (l >> 4) + 4 r][col = l & 15]
}

// ---- 2. issue interval / dependent latency -----------------------------------------------------------------------------------
template <int NACC>
__global__ void k_rate(double* out, long long* t, int n) {
  double4_t c[NACC];
#pragma unroll
  for (int m = 0; m < NACC; ++m) c[m] = double4_t{0.0, 0.0, 0.0, 0.0};
  const double a = 1e-3 * threadIdx.x, b = 1e-3 * (threadIdx.x % 7);
  __syncthreads();
  const long long c0 = clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int m = 0; m < NACC; ++m) c[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[m], 0, 0, 0);
  }
  const long long c1 = clock64();
  double s = 0;
#pragma unroll
  for (int m = 0; m < NACC; ++m) s += c[m][0] + c[m][1] + c[m][2] + c[m][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}

// ---- 3. ring-tiled window step ------------------------------------------------------------------------------------------------
// NT x NT tiles of 16 x 16 (upper triangle incl. diagonal: NT (NT + 1) / 2 tiles), dealt round-robin to NW waves; tiles per wave TW.
template <int NT, int NW, int K>
__global__ void __launch_bounds__(64 * NW) k_window(double* out, long long* t, int n) {
  constexpr int W = 16 * NT, LD = W + 4, NTILE = NT * (NT + 1) / 2, TW = (NTILE + NW - 1) / NW, KG = K / 4;
  __shared__ __attribute__((aligned(16))) double X[2][K * LD];
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
  for (int e = tid; e < 2 * K * LD; e += blockDim.x) (&X[0][0])[e] = 1e-3 * (e % 29);
  // tile list of this wave: tile index q = wave + NW * m  ->  (I, J), I <= J
  int tI[TW], tJ[TW];
  bool ok[TW];
#pragma unroll
  for (int m = 0; m < TW; ++m) {
    int q = wave + NW * m;
    ok[m] = q < NTILE;
    if (!ok[m]) q = 0;
    int I = 0;
    while (q >= NT - I) q -= NT - I, ++I;
    tI[m] = I, tJ[m] = I + q;
  }
  double4_t acc[TW];
#pragma unroll
  for (int m = 0; m < TW; ++m) acc[m] = double4_t{1.0 * m, 2.0, 3.0, 4.0};
  __syncthreads();
  const long long c0 = clock64();
  for (int it = 0; it < n; ++it) {
    const double* x = X[it & 1];
#pragma unroll
    for (int m = 0; m < TW; ++m) {
      if (!ok[m]) continue;
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const double a = -x[((l >> 4) + 4 * g) * LD + 16 * tI[m] + (l & 15)];
        const double b = x[((l >> 4) + 4 * g) * LD + 16 * tJ[m] + (l & 15)];
        acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
      }
    }
    // the wave that owns the pivot tile row publishes K rows of it for the next panel (2 accumulator registers per tile)
    if (wave == (it % NW)) {
      double* dst = &X[(it + 1) & 1][0];
#pragma unroll
      for (int m = 0; m < TW; ++m)
        if (ok[m] && (l >> 4) < K / 2) dst[(l >> 4) * LD + 16 * tJ[m] + (l & 15)] = 1e-9 * acc[m][0], dst[((l >> 4) + 4) * LD + 16 * tJ[m] + (l & 15)] = 1e-9 * acc[m][1];
    }
    lds_barrier();
  }
  const long long c1 = clock64();
  double s = 0;
#pragma unroll
  for (int m = 0; m < TW; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[tid] = s;
  if (tid == 0) t[0] = c1 - c0;
}

// ---- 4. panel chain of a B x B pivot block in one wave ------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ int uidx(int a, int c) { return a * B - a * (a - 1) / 2 + (c - a); }

template <int B, int MODE>
__global__ void __launch_bounds__(64) k_panel(double* out, long long* t, int n) {
  __shared__ double lds[B * B + 2 * B * 64];
  for (int e = threadIdx.x; e < B * B; e += 64) lds[e] = (e / B == e % B ? 8.0 : 0.0) + 0.3 / (1 + e / B + e % B);
  for (int e = threadIdx.x; e < 2 * B * 64; e += 64) lds[B * B + e] = 0.01 * (e % 13);
  __syncthreads();
  constexpr int NU = B * (B + 1) / 2;
  double carry = 0.0, accum = 0.0;
  const long long c0 = clock64();
  for (int it = 0; it < n; ++it) {
    double U[NU], inv[B], v[2][B], x[2][B];
    {
      int p = 0;
#pragma unroll
      for (int a = 0; a < B; ++a)
#pragma unroll
        for (int c = a; c < B; ++c) U[p++] = lds[B * a + c] + carry;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int a = 0; a < B; ++a) v[m][a] = lds[B * B + (2 * a + m) * 64 + threadIdx.x];
#pragma unroll
    for (int a = 0; a < B; ++a) {
      double d = U[uidx<B>(a, a)];
      if (MODE == 0) {  // left-looking pivot: the whole sum sits on the chain
#pragma unroll
        for (int k = 0; k < a; ++k) d = fma(-U[uidx<B>(k, a)], U[uidx<B>(k, a)], d);
      }
      const double y = __builtin_amdgcn_rsq(d);
      const double e = fma(-d * y, y, 1.0);
      const double rs = fma(y * e, fma(0.375, e, 0.5), y);
      inv[a] = rs;
      const double nrs = -rs;
#pragma unroll
      for (int c = a + 1; c < B; ++c) {
        double tt = U[uidx<B>(a, c)];
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < a; ++k) tt = fma(-U[uidx<B>(k, a)], U[uidx<B>(k, c)], tt);
        }
        U[uidx<B>(a, c)] = tt * (MODE == 0 ? rs : nrs);
      }
      if (MODE == 1) {
        // right-looking: row a is final -> update the trailing block and the right-hand sides now (short chains: one FMA deep per
        // pivot on the diagonal), the column solves ride along with the factorisation. Off-diagonal entries are kept negated.
#pragma unroll
        for (int r = a + 1; r < B; ++r)
#pragma unroll
          for (int c = r; c < B; ++c) U[uidx<B>(r, c)] = fma(-U[uidx<B>(a, r)], U[uidx<B>(a, c)], U[uidx<B>(r, c)]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          x[m][a] = v[m][a] * rs;
#pragma unroll
          for (int r = a + 1; r < B; ++r) v[m][r] = fma(U[uidx<B>(a, r)], x[m][a], v[m][r]);
        }
      }
    }
    if (MODE == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int a = 0; a < B; ++a) {
          double tt = v[m][a];
#pragma unroll
          for (int k = 0; k < a; ++k) tt = fma(-U[uidx<B>(k, a)], x[m][k], tt);
          x[m][a] = tt * inv[a];
        }
    }
    carry = (x[0][B - 1] + x[1][B - 1]) * 1e-30;
    accum += x[0][0] + x[1][B / 2];
  }
  const long long c1 = clock64();
  out[threadIdx.x] = accum;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}

template <class F>
static double run2(F&& launch, long long* t) {
  launch(), (void)hipDeviceSynchronize();
  launch(), (void)hipDeviceSynchronize();
  return double(t[0]);
}

int main() {
  double *out, *dA, *dB, *dD;
  long long* t;
  (void)hipMalloc(&out, 1 << 16), (void)hipMallocManaged(&t, 64);
  (void)hipMallocManaged(&dA, 64 * 8), (void)hipMallocManaged(&dB, 64 * 8), (void)hipMallocManaged(&dD, 256 * 8);
  // 1. layout
  {
    for (int i = 0; i < 16; ++i)
      for (int k = 0; k < 4; ++k) dA[i * 4 + k] = (i == k) ? 1.0 : 0.0;  // A = [I_4 ; 0]
    for (int k = 0; k < 4; ++k)
      for (int j = 0; j < 16; ++j) dB[k * 16 + j] = 100.0 * k + j;  // asymmetric
    k_layout<<<1, 64>>>(dA, dB, dD);
    (void)hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) bad += dD[i * 16 + j] != (i < 4 ? 100.0 * i + j : 0.0);
    std::vector<double> ref(256, 0.0);
    for (int e = 0; e < 64; ++e) dA[e] = std::sin(1.0 + e), dB[e] = std::cos(0.3 * e);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j)
        for (int k = 0; k < 4; ++k) ref[i * 16 + j] = std::fma(dA[i * 4 + k], dB[k * 16 + j], ref[i * 16 + j]);
    k_layout<<<1, 64>>>(dA, dB, dD);
    (void)hipDeviceSynchronize();
    double err = 0;
    for (int e = 0; e < 256; ++e) err = std::fmax(err, std::fabs(dD[e] - ref[e]));
    printf("layout: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[(l>>4)+4r][l&15]: %s (identity check %d wrong, random product max err %.3g)\n",
           bad == 0 && err < 1e-14 ? "CONFIRMED" : "WRONG", bad, err);
  }
  const int n = 4000;
  // 2. rate
  for (int threads : {64, 256, 512}) {
    const double d1 = run2([&] { k_rate<1><<<1, threads>>>(out, t, n); }, t) / n;
    const double d4 = run2([&] { k_rate<4><<<1, threads>>>(out, t, n); }, t) / (4.0 * n);
    const double d8 = run2([&] { k_rate<8><<<1, threads>>>(out, t, n); }, t) / (8.0 * n);
    printf("mfma_f64_16x16x4 %3d threads: dependent %.1f clk, 4 independent %.1f clk each, 8 independent %.1f clk each\n", threads, d1, d4, d8);
  }
  // 3. window steps
  {
    const int ns = 2000;
    printf("ring window, rank-8 update per step (2 MFMA + 4 ds_read_b64 per tile), 1 LDS barrier per step:\n");
    printf("  W =  96 (21 tiles) on 2 waves: %.0f clk / step\n", run2([&] { k_window<6, 2, 8><<<1, 128>>>(out, t, ns); }, t) / ns);
    printf("  W =  96 (21 tiles) on 4 waves: %.0f clk / step\n", run2([&] { k_window<6, 4, 8><<<1, 256>>>(out, t, ns); }, t) / ns);
    printf("  W =  96 (21 tiles) on 8 waves: %.0f clk / step\n", run2([&] { k_window<6, 8, 8><<<1, 512>>>(out, t, ns); }, t) / ns);
    printf("  W = 144 (45 tiles) on 4 waves: %.0f clk / step\n", run2([&] { k_window<9, 4, 8><<<1, 256>>>(out, t, ns); }, t) / ns);
    printf("  W = 144 (45 tiles) on 8 waves: %.0f clk / step\n", run2([&] { k_window<9, 8, 8><<<1, 512>>>(out, t, ns); }, t) / ns);
    printf("  W = 256 (136 tiles) on 8 waves: %.0f clk / step\n", run2([&] { k_window<16, 8, 8><<<1, 512>>>(out, t, ns); }, t) / ns);
    printf("  W = 256 (136 tiles) on 16 waves: %.0f clk / step\n", run2([&] { k_window<16, 16, 8><<<1, 1024>>>(out, t, ns); }, t) / ns);
    printf("  W =  96, rank-16 (4 MFMA per tile) on 4 waves: %.0f clk / step\n", run2([&] { k_window<6, 4, 16><<<1, 256>>>(out, t, ns); }, t) / ns);
    printf("  W = 112, rank-16 (28 tiles) on 4 waves: %.0f clk / step\n", run2([&] { k_window<7, 4, 16><<<1, 256>>>(out, t, ns); }, t) / ns);
  }
  // 4. panel chains
  {
    const int np = 2000;
    printf("panel chain (redundant register Cholesky + 2 column solves per lane, one wave):\n");
    printf("  6 x 6  plain        : %.0f clk\n", run2([&] { k_panel<6, 0><<<1, 64>>>(out, t, np); }, t) / np);
    printf("  6 x 6  interleaved  : %.0f clk\n", run2([&] { k_panel<6, 1><<<1, 64>>>(out, t, np); }, t) / np);
    printf("  8 x 8  plain        : %.0f clk\n", run2([&] { k_panel<8, 0><<<1, 64>>>(out, t, np); }, t) / np);
    printf("  8 x 8  interleaved  : %.0f clk\n", run2([&] { k_panel<8, 1><<<1, 64>>>(out, t, np); }, t) / np);
    printf("  12 x 12 interleaved : %.0f clk\n", run2([&] { k_panel<12, 1><<<1, 64>>>(out, t, np); }, t) / np);
    printf("  16 x 16 interleaved : %.0f clk\n", run2([&] { k_panel<16, 1><<<1, 64>>>(out, t, np); }, t) / np);
  }
  return 0;
}

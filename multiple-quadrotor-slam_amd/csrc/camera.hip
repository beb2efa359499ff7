// Streaming camera-model kernels either side of triangulation:
// undistort + normalise pixel observations (cv2.undistortPoints at slam2.py:551-552) and project +
// RMS reprojection error (calibration_tools.reprojection_error :116-124, slam2.py:562-563).
// HBM-bound: 16 B in / 16 B out (undistort), 24 (+16) B in / 16 B out (project) per point.
#include "mqs_common.h"
#include "cam_math.h"

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void undistort_kernel(const double *__restrict__ pix, const double *__restrict__ intr,
                                                           int64_t N, double *__restrict__ out)
{
    __shared__ double sI[9];
    if (threadIdx.x < 9) sI[threadIdx.x] = intr[threadIdx.x];
    __syncthreads();
    const double2 *p2 = reinterpret_cast<const double2 *>(pix);
    double2 *o2 = reinterpret_cast<double2 *>(out);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
        const double2 p = p2[i];
        double x, y;
        mqs::cam::undistort_pixel(sI, p.x, p.y, x, y);
        o2[i] = make_double2(x, y);
    }
}

// uv_out (may be null), depth_out (may be null); when imgp != null accumulates sum |uv - imgp|^2
// into partials[blockIdx] (fixed-order, reproducible).
__global__ __launch_bounds__(kBlock) void project_kernel(const double *__restrict__ points, const double *__restrict__ P,
                                                         const double *__restrict__ intr, const double *__restrict__ imgp,
                                                         int64_t N, double *__restrict__ uv_out,
                                                         double *__restrict__ depth_out, double *__restrict__ partials)
{
    __shared__ double sP[12], sI[9], sX[kBlock * 3], sR[kBlock / 64];
    const int tid = threadIdx.x;
    if (tid < 12) sP[tid] = P[tid];
    if (tid < 9) sI[tid] = intr[tid];
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < N; base += stride) {
        const int64_t rem = N - base;
        const int npts = rem < kBlock ? (int)rem : kBlock;
        const int ndbl = npts * 3;
        __syncthreads();
        const double2 *src = reinterpret_cast<const double2 *>(points + base * 3);
        double2 *dst = reinterpret_cast<double2 *>(sX);
        for (int p = tid; p < (ndbl >> 1); p += kBlock) dst[p] = src[p];
        if ((ndbl & 1) && tid == 0) sX[ndbl - 1] = points[base * 3 + ndbl - 1];
        __syncthreads();
        if (tid < npts) {
            const int64_t i = base + tid;
            double u, v;
            const double Z = mqs::cam::project(sP, sI, sX[3 * tid], sX[3 * tid + 1], sX[3 * tid + 2], u, v);
            if (uv_out) reinterpret_cast<double2 *>(uv_out)[i] = make_double2(u, v);
            if (depth_out) depth_out[i] = Z;
            if (imgp) {
                const double2 m = reinterpret_cast<const double2 *>(imgp)[i];
                const double du = u - m.x, dv = v - m.y;
                acc += du * du + dv * dv;
            }
        }
    }
    if (partials) {
#pragma unroll
        for (int h = 32; h >= 1; h >>= 1) acc += __shfl_xor(acc, h);
        if ((tid & 63) == 0) sR[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) partials[blockIdx.x] = sR[0] + sR[1] + sR[2] + sR[3];
    }
}

__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const double *__restrict__ partials, int n, double *__restrict__ out)
{
    __shared__ double s[kBlock];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) a += partials[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int h = kBlock / 2; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) s[threadIdx.x] += s[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

constexpr int kMaxPartials = 2048;

}  // namespace

extern "C" {

int mqs_undistort_points_dev(const double *pixels, const double *intr, int64_t N, double *out, void *stream)
{
    MQS_ARG_CHECK(N >= 0, "N >= 0");
    if (N == 0) return MQS_OK;
    MQS_ARG_CHECK(pixels && intr && out, "pointers must not be null");
    MQS_ARG_CHECK(mqs_aligned16(pixels) && mqs_aligned16(out), "device pointers must be 16-byte aligned");
    hipLaunchKernelGGL(undistort_kernel, dim3(mqs_stream_grid(N, kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       pixels, intr, N, out);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int64_t mqs_project_workspace_bytes(void) { return (kMaxPartials + 1) * (int64_t)sizeof(double); }

int mqs_project_points_dev(const double *points, const double *P, const double *intr, const double *imgp, int64_t N,
                           double *uv_out, double *depth_out, double *sqerr_out, void *workspace, int64_t workspace_bytes,
                           void *stream_)
{
    MQS_ARG_CHECK(N >= 0, "N >= 0");
    MQS_ARG_CHECK(P && intr && (N == 0 || points), "pointers must not be null");
    MQS_ARG_CHECK(!sqerr_out || (imgp && workspace && workspace_bytes >= mqs_project_workspace_bytes()),
                  "sqerr_out needs imgp and a workspace of mqs_project_workspace_bytes()");
    MQS_ARG_CHECK(mqs_aligned16(points) && mqs_aligned16(uv_out) && mqs_aligned16(imgp), "device pointers must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int64_t g = (N + kBlock - 1) / kBlock;
    const int grid = (int)(g < 1 ? 1 : (g > kMaxPartials ? kMaxPartials : g));
    double *partials = sqerr_out ? static_cast<double *>(workspace) : nullptr;
    hipLaunchKernelGGL(project_kernel, dim3(grid), dim3(kBlock), 0, stream, points, P, intr, imgp, N, uv_out, depth_out,
                       partials);
    if (sqerr_out) hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, stream, partials, grid, sqerr_out);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// host-pointer wrappers
int mqs_undistort_points(mqs_ctx *ctx, const double *pixels, const double *intr, int64_t N, double *out)
{
    MQS_ARG_CHECK(ctx != nullptr && N >= 0, "ctx not null, N >= 0");
    if (N == 0) return MQS_OK;
    MQS_ARG_CHECK(pixels && intr && out, "pointers must not be null");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_p = 0, o_o = up((size_t)N * 16), o_i = up(o_o + (size_t)N * 16), total = up(o_i + 72);
    mqs_stage st;
    int rc = mqs_stage_begin(ctx, total, &st);
    if (rc != MQS_OK) return rc;
    char *d = st.base;
    MQS_HIP_CHECK(mqs_stage_in(&st, d + o_p, pixels, (size_t)N * 16));
    MQS_HIP_CHECK(mqs_stage_in(&st, d + o_i, intr, 72));
    rc = mqs_undistort_points_dev((const double *)(d + o_p), (const double *)(d + o_i), N, (double *)(d + o_o), ctx->stream);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(mqs_stage_out(&st, out, d + o_o, (size_t)N * 16));
    MQS_HIP_CHECK(mqs_stage_end(&st));
    return MQS_OK;
}

int mqs_project_points(mqs_ctx *ctx, const double *points, const double *P, const double *intr, const double *imgp,
                       int64_t N, double *uv_out, double *depth_out, double *sqerr_out)
{
    MQS_ARG_CHECK(ctx != nullptr && N >= 0, "ctx not null, N >= 0");
    MQS_ARG_CHECK(P && intr && (N == 0 || points), "pointers must not be null");
    MQS_ARG_CHECK(!sqerr_out || imgp, "sqerr_out needs imgp");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = up(off + b); return o; };
    const size_t o_x = take((size_t)N * 24), o_m = take((size_t)N * 16), o_uv = take((size_t)N * 16), o_z = take((size_t)N * 8);
    const size_t o_P = take(96), o_i = take(72), o_e = take(8), o_w = take((size_t)mqs_project_workspace_bytes());
    int rc = mqs_ctx_reserve(ctx, off);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    hipStream_t s = ctx->stream;
    // small problems: one packed copy in (points, measurements, P, intrinsics), one copy out (uv, depth, error)
    mqs_stage st;
    st.zero_copy = false;
    if (o_w <= kZeroCopyMax && N > 0) {
        rc = mqs_stage_begin(ctx, o_w, &st);
        if (rc != MQS_OK) return rc;
    }
    if (st.zero_copy) {
        char *h = st.base;
        memcpy(h + o_x, points, (size_t)N * 24);
        if (imgp) memcpy(h + o_m, imgp, (size_t)N * 16);
        memcpy(h + o_P, P, 96);
        memcpy(h + o_i, intr, 72);
        MQS_HIP_CHECK(hipMemcpyAsync(d, h, o_e, hipMemcpyHostToDevice, s));
        rc = mqs_project_points_dev((const double *)(d + o_x), (const double *)(d + o_P), (const double *)(d + o_i),
                                    imgp ? (const double *)(d + o_m) : nullptr, N, uv_out ? (double *)(d + o_uv) : nullptr,
                                    depth_out ? (double *)(d + o_z) : nullptr, sqerr_out ? (double *)(d + o_e) : nullptr, d + o_w,
                                    mqs_project_workspace_bytes(), s);
        if (rc != MQS_OK) return rc;
        MQS_HIP_CHECK(hipMemcpyAsync(h + o_uv, d + o_uv, o_w - o_uv, hipMemcpyDeviceToHost, s));
        MQS_HIP_CHECK(hipStreamSynchronize(s));
        if (uv_out) memcpy(uv_out, h + o_uv, (size_t)N * 16);
        if (depth_out) memcpy(depth_out, h + o_z, (size_t)N * 8);
        if (sqerr_out) memcpy(sqerr_out, h + o_e, 8);
        return MQS_OK;
    }
    if (N > 0) MQS_HIP_CHECK(hipMemcpyAsync(d + o_x, points, (size_t)N * 24, hipMemcpyHostToDevice, s));
    if (N > 0 && imgp) MQS_HIP_CHECK(hipMemcpyAsync(d + o_m, imgp, (size_t)N * 16, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_P, P, 96, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_i, intr, 72, hipMemcpyHostToDevice, s));
    rc = mqs_project_points_dev((const double *)(d + o_x), (const double *)(d + o_P), (const double *)(d + o_i),
                                imgp ? (const double *)(d + o_m) : nullptr, N, uv_out ? (double *)(d + o_uv) : nullptr,
                                depth_out ? (double *)(d + o_z) : nullptr, sqerr_out ? (double *)(d + o_e) : nullptr, d + o_w,
                                mqs_project_workspace_bytes(), s);
    if (rc != MQS_OK) return rc;
    if (N > 0 && uv_out) MQS_HIP_CHECK(hipMemcpyAsync(uv_out, d + o_uv, (size_t)N * 16, hipMemcpyDeviceToHost, s));
    if (N > 0 && depth_out) MQS_HIP_CHECK(hipMemcpyAsync(depth_out, d + o_z, (size_t)N * 8, hipMemcpyDeviceToHost, s));
    if (sqerr_out) MQS_HIP_CHECK(hipMemcpyAsync(sqerr_out, d + o_e, 8, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipStreamSynchronize(s));
    return MQS_OK;
}

}  // extern "C"

// select_top_k / knn_point of lib/utils/tf_ops/grouping (tf_grouping.py:103-160):
//   selection_sort_gpu (tf_grouping_g.cu:404-443): for every row of dist [b,m,n] the k smallest entries are moved to the
//   front by k steps of selection sort -- step s finds the first minimum of positions s..n-1 (position s itself wins
//   ties) and swaps it with position s; values AND the index permutation outi are returned in full ([b,m,n]), so
//   the arrangement of the tail after the swaps is part of the result.
// The reference runs one THREAD per row.  Here a row is owned by one 256-thread workgroup: the row and its index
// permutation sit in LDS (n <= 16384: 128 KiB), a step is a strided scan + a wave/workgroup arg-min whose tie rule
// (lowest position) reproduces the serial scan, and one thread performs the swap.
// knn_point builds dist as reduce_sum((xyz1 - xyz2)^2, -1) with TensorFlow ops (tf_grouping.py:146-150): squares
// rounded separately, summed over the channels ascending (pairwise_sqdist_kernel).
#include "sa_common.h"

namespace {

constexpr int kSelBlock = 256;
constexpr int kSelMaxN = 16384;

__global__ __launch_bounds__(kSelBlock) void selection_sort_kernel(int n, int k, const float *__restrict__ dist,
                                                                   int *__restrict__ outi, float *__restrict__ out) {
    extern __shared__ float s_dyn[];
    float *s_v = s_dyn;
    int *s_i = (int *)(s_dyn + n);
    __shared__ float s_wv[kSelBlock / 64];
    __shared__ int s_wp[kSelBlock / 64];
    const size_t row = (size_t)blockIdx.x * n;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int j = t; j < n; j += kSelBlock) {
        s_v[j] = dist[row + j];
        s_i[j] = j;
    }
    __syncthreads();
    const int steps = k < n ? k : n;
    for (int s = 0; s < steps; ++s) {
        // candidates s..n-1; a thread (wave) without one carries (+inf, INT_MAX) and loses every tie on position
        float bv = __builtin_inff();
        unsigned bp = 0x7FFFFFFFu;
        for (int j = s + t; j < n; j += kSelBlock) {                 // ascending positions per thread: strict <
            const float v = s_v[j];
            if (bp == 0x7FFFFFFFu || v < bv) { bv = v; bp = (unsigned)j; }
        }
        const float wmin = -sa::wave_allmax(-bv);
        const unsigned pos = sa::wave_allmin_u32(bv == wmin ? bp : 0xFFFFFFFFu);
        if (lane == 0) { s_wv[w] = wmin; s_wp[w] = (int)pos; }
        __syncthreads();
        if (t == 0) {
            float mv = s_wv[0];
            int mp = s_wp[0];
            for (int i = 1; i < kSelBlock / 64; ++i)
                if (s_wv[i] < mv || (s_wv[i] == mv && s_wp[i] < mp)) { mv = s_wv[i]; mp = s_wp[i]; }
            if (mp != s) {                                               // tf_grouping_g.cu:432-439
                const float tv = s_v[mp]; s_v[mp] = s_v[s]; s_v[s] = tv;
                const int ti = s_i[mp]; s_i[mp] = s_i[s]; s_i[s] = ti;
            }
        }
        __syncthreads();
    }
    for (int j = t; j < n; j += kSelBlock) {
        out[row + j] = s_v[j];
        outi[row + j] = s_i[j];
    }
}

// dist[b, j, i] = sum_l (xyz1[b,i,l] - xyz2[b,j,l])^2, products rounded, channels ascending
__global__ __launch_bounds__(256) void pairwise_sqdist_kernel(int n, int m, int c, const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2, float *__restrict__ dist) {
    const int bi = blockIdx.z, j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *a = xyz1 + ((size_t)bi * n + i) * c;
    const float *q = xyz2 + ((size_t)bi * m + j) * c;
    float d = 0.0f;
    for (int l = 0; l < c; ++l) {
        const float df = a[l] - q[l];
        const float sq = df * df;
        d = l == 0 ? sq : d + sq;
    }
    dist[((size_t)bi * m + j) * n + i] = d;
}

}  // namespace

// selectionSortLauncher(b,n,m,k,dist,outi,out) -- tf_grouping.cpp:411.  dist/outi/out [b,m,n].
extern "C" int sa_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out,
                                 hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || k <= 0 || !dist || !outi || !out) return SA_ERR_INVALID;
    if (n > kSelMaxN || (long)b * m > 0x7FFFFFFF) return SA_ERR_UNSUPPORTED;
    const size_t lds = (size_t)n * 8;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)selection_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(selection_sort_kernel, dim3((unsigned)(b * m)), dim3(kSelBlock), lds, stream, n, k, dist, outi, out);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// the distance matrix of knn_point (tf_grouping.py:146-150): xyz1 [b,n,c] dataset, xyz2 [b,m,c] queries -> [b,m,n]
extern "C" int sa_pairwise_sqdist(int b, int n, int m, int c, const float *xyz1, const float *xyz2, float *dist,
                                  hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || !xyz1 || !xyz2 || !dist) return SA_ERR_INVALID;
    if (m > 65535 || b > 65535) return SA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pairwise_sqdist_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)m, (unsigned)b), dim3(256), 0,
                       stream, n, m, c, xyz1, xyz2, dist);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// eval_kernels.hip -- the O(N^2 log N) part of CoverAlgorithm.getEvalStatistics (CoverAlgorithm.py:362-390) on
// the GPU: for every song, the 1-based ranks of its clique mates in the descending-score order of its row of
// the similarity matrix, sorted ascending.  The reference argsorts every row (np.argsort(-D, 1), :362) and looks
// the mates up (:381); a rank is just a count, so no sort is needed:
//     rank_i(j) = 1 + #{k != i : D[i][k] > D[i][j]  or  (D[i][k] == D[i][j] and k < j)}.
// The mean rank, MRR, median rank, MAP and Top-X then follow on the host from these few numbers exactly as in
// the reference (:386-402).  Ties: np.argsort's default sort is not stable, so the reference's order among equal
// scores is an accident of introsort; here equal scores rank in song-index order.  On tie-free rows the ranks
// are identical.
#include "common.h"
#include "wave_ops.h"

namespace acoss {

constexpr int EVAL_THREADS = 256;
constexpr int EVAL_MAX_MATES = 4096;

__global__ __launch_bounds__(EVAL_THREADS) void eval_ranks_kernel(const float *__restrict__ D, int N, int64_t pitch,
                                                                 const int32_t *__restrict__ clique_id,
                                                                 const int64_t *__restrict__ mate_off,
                                                                 int32_t *__restrict__ mate_ranks)
{
    __shared__ int mate[EVAL_MAX_MATES];
    __shared__ int rank[EVAL_MAX_MATES];
    __shared__ int n_mates;
    __shared__ int wave_cnt[EVAL_THREADS / 64];
    const int i = blockIdx.x;
    const int want = (int)(mate_off[i + 1] - mate_off[i]);
    if (want <= 0) return;
    const int cid = clique_id[i];
    const float *row = D + (int64_t)i * pitch;
    if (threadIdx.x == 0) n_mates = 0;
    __syncthreads();
    // the mates, in index order (stable compaction: one pass per 256 songs, ballot prefix inside each wave)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j0 = 0; j0 < N; j0 += EVAL_THREADS) {
        const int j = j0 + threadIdx.x;
        const bool is = j < N && j != i && clique_id[j] == cid;
        const unsigned long long m = __ballot(is);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = n_mates;
        for (int v = 0; v < wave; v++) before += wave_cnt[v];
        const int slot = before + __popcll(m & ((1ull << lane) - 1ull));
        if (is && slot < EVAL_MAX_MATES) mate[slot] = j;
        __syncthreads();
        if (threadIdx.x == 0) n_mates += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int K = min(min(n_mates, want), EVAL_MAX_MATES);
    // ranks: wave w counts for mates w, w + 4, ...; the row stays in L2 between passes
    for (int t = wave; t < K; t += EVAL_THREADS / 64) {
        const int j = mate[t];
        const float v = row[j];
        int c = 0;
        for (int k = lane; k < N; k += 64) {
            const float x = row[k];
            c += (k != i) & ((x > v) | ((x == v) & (k < j)));
        }
        c = wave_sum(c);
        if (lane == 0) rank[t] = c + 1;
    }
    __syncthreads();
    // ascending order: ranks are distinct (a strict total order), so a rank's place is the number of smaller ones
    int32_t *out = mate_ranks + mate_off[i];
    for (int t = threadIdx.x; t < K; t += EVAL_THREADS) {
        const int r = rank[t];
        int place = 0;
        for (int u = 0; u < K; u++) place += rank[u] < r;
        out[place] = r;
    }
}

}  // namespace acoss

using namespace acoss;

extern "C" {

int acoss_eval_ranks(const float *D, int N, int64_t row_pitch, const int32_t *clique_id, const int64_t *mate_off,
                     int max_mates, int32_t *mate_ranks, void *stream)
{
    if (!D || !clique_id || !mate_off || !mate_ranks || N < 1 || row_pitch < N || max_mates < 0) {
        set_error("eval_ranks: bad argument");
        return ACOSS_EINVAL;
    }
    if (max_mates > EVAL_MAX_MATES) {
        set_error("eval_ranks: cliques of more than %d songs are not supported", EVAL_MAX_MATES + 1);
        return ACOSS_ENOTSUP;
    }
    hipLaunchKernelGGL(eval_ranks_kernel, dim3((unsigned)N), dim3(EVAL_THREADS), 0, (hipStream_t)stream, D, N, row_pitch,
                       clique_id, mate_off, mate_ranks);
    return launch_check("eval_ranks_kernel");
}

}  // extern "C"

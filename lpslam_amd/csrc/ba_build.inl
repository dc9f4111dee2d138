// ba_build.inl -- structure phase of a bundle-adjustment problem on the device (included by ba.hip).
//
// [UPSTREAM] g2o BlockSolver::buildStructure + the index bookkeeping of OpenVSLAM's local / global bundle adjuster, which the
// reference runs once per keyframe on its mapping thread (/root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:239).
// The caller's observation list (array of lpslam_hip_ba_obs, any order) becomes, without a host sort:
//   storage order   observations keyframe by keyframe and, inside a keyframe, by landmark (ties: caller order)  -> o_* arrays, o_orig
//   ps_start        first observation of every keyframe
//   pt_start/pt_obs CSR by landmark, entries ascending (= keyframe order)
//   blk_start/terms per pose-block pair (slot a <= slot c) the list of (observation of a, observation of c, landmark), landmark order
// The tool is a dense keyframe x landmark table: A[p][j] = number of observations of landmark j in keyframe p (1 almost always),
// R[p][j] = exclusive scan of A along j.  The storage position of an observation is ps_start[p] + R[p][j] (+ its rank among
// duplicates); the partner of an observation in another keyframe c is found by one look-up, A[c][j] / R[c][j], so a pair list is a
// filtered copy of keyframe a's observation range.  200 keyframes x 30 000 landmarks are 2 x 24 MB of the 288 GB.
// Every step is order-free or ordered by construction: no result depends on the arrival order of an atomic.

constexpr int BS_THREADS = 1024;

// exclusive scan over the workgroup (1024 threads): returns the prefix of `x`, *total receives the sum
__device__ __forceinline__ int bs_block_scan(int x, int* total)
{
    __shared__ int s_wave[BS_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
    __syncthreads();                                   // s_wave may still be read by the previous call
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < BS_THREADS / 64; ++w) { const int t = s_wave[w]; if (w < wave) base += t; sum += t; }
    *total = sum;
    return base + incl - x;
}

// A[p][j] += 1 per observation; landmark degrees
__global__ __launch_bounds__(256) void k_bs_count(const lpslam_hip_ba_obs* __restrict__ obs, int n_obs, int n_points, int* A, int* pt_count)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_obs) return;
    const int p = obs[k].pose, j = obs[k].point;
    atomicAdd(&A[(size_t)p * n_points + j], 1);
    atomicAdd(&pt_count[j], 1);
}

// R[p][.] = exclusive scan of A[p][.] (one workgroup per keyframe); ps_count[p] = observations of keyframe p
__global__ __launch_bounds__(BS_THREADS) void k_bs_rowscan(const int* __restrict__ A, int* R, int n_points, int* ps_count)
{
    const int p = blockIdx.x;
    const int* a = A + (size_t)p * n_points;
    int* r = R + (size_t)p * n_points;
    int carry = 0;
    for (int j0 = 0; j0 < n_points; j0 += 4 * BS_THREADS) {
        const int j = j0 + 4 * (int)threadIdx.x;
        int x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = j + u < n_points ? a[j + u] : 0;
        int tot;
        int pre = carry + bs_block_scan(x[0] + x[1] + x[2] + x[3], &tot);
#pragma unroll
        for (int u = 0; u < 4; ++u) { if (j + u < n_points) r[j + u] = pre; pre += x[u]; }
        carry += tot;
    }
    if (threadIdx.x == 0) ps_count[p] = carry;
}

// workgroup 0: ps_start = exclusive scan of ps_count (n_poses + 1 entries); workgroup 1: pt_start from pt_count; workgroup 2
// (with blk_count != nullptr): blk_start from blk_count, in place
__global__ __launch_bounds__(BS_THREADS) void k_bs_starts(const int* ps_count, int* ps_start, int n_poses, const int* pt_count, int* pt_start, int n_points)
{
    const int* in = blockIdx.x == 0 ? ps_count : pt_count;
    int* out = blockIdx.x == 0 ? ps_start : pt_start;
    const int n = blockIdx.x == 0 ? n_poses : n_points;
    int carry = 0;
    for (int i0 = 0; i0 < n; i0 += BS_THREADS) {
        const int i = i0 + (int)threadIdx.x;
        const int x = i < n ? in[i] : 0;
        int tot;
        const int pre = carry + bs_block_scan(x, &tot);
        if (i < n) out[i] = pre;
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// first pass of the placement: every observation takes a slot of its (keyframe, landmark) group; inside a group of duplicates
// the slot is arbitrary here (an atomic on the high half of A) and put in caller order by k_bs_gather
__global__ __launch_bounds__(256) void k_bs_scatter(const lpslam_hip_ba_obs* __restrict__ obs, int n_obs, int n_points, int* A, const int* __restrict__ R,
                                                    const int* __restrict__ ps_start, int* slot_of)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_obs) return;
    const int p = obs[k].pose, j = obs[k].point;
    const size_t e = (size_t)p * n_points + j;
    const int base = ps_start[p] + R[e];
    const int c = A[e] & 0xFFFF;
    const int d = c == 1 ? 0 : (atomicAdd(&A[e], 0x10000) >> 16);
    slot_of[base + d] = k;
}

// second pass: storage position s takes the (s - first)-th smallest caller index of its group and copies that observation into
// the SoA arrays
__global__ __launch_bounds__(256) void k_bs_gather(const lpslam_hip_ba_obs* __restrict__ obs, int n_obs, int n_points, const int* __restrict__ A,
                                                   const int* __restrict__ R, const int* __restrict__ ps_start, const int* __restrict__ slot_of,
                                                   int* o_orig, int* o_pose, int* o_point, double* o_u, double* o_v, double* o_ur, double* o_w, uint8_t* o_active,
                                                   uint8_t* act_in)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_obs) return;
    int k = slot_of[s];
    const int p = obs[k].pose, j = obs[k].point;
    const size_t e = (size_t)p * n_points + j;
    const int c = A[e] & 0xFFFF;
    if (c > 1) {
        const int first = ps_start[p] + R[e], want = s - first;
        // rank selection among the c caller indices of the group (c is tiny; duplicates of one landmark in one keyframe are rare)
        for (int a = 0; a < c; ++a) {
            const int ka = slot_of[first + a];
            int rank = 0;
            for (int b2 = 0; b2 < c; ++b2) rank += slot_of[first + b2] < ka ? 1 : 0;
            if (rank == want) k = ka;
        }
    }
    const lpslam_hip_ba_obs o = obs[k];
    o_orig[s] = k; o_pose[s] = o.pose; o_point[s] = o.point;
    o_u[s] = o.u; o_v[s] = o.v; o_ur[s] = o.ur; o_w[s] = o.inv_sigma2;
    o_active[s] = 1; act_in[s] = 1;
}

// CSR by landmark: thread per landmark walks the keyframes in order (column of A / R)
__global__ __launch_bounds__(256) void k_bs_ptfill(const int* __restrict__ A, const int* __restrict__ R, int n_poses, int n_points, const int* __restrict__ ps_start,
                                                   const int* __restrict__ pt_start, int* pt_obs)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_points) return;
    int t = pt_start[j];
    for (int p = 0; p < n_poses; ++p) {
        const size_t e = (size_t)p * n_points + j;
        const int c = A[e] & 0xFFFF;
        const int first = ps_start[p] + R[e];
        for (int d = 0; d < c; ++d) pt_obs[t++] = first + d;
    }
}

// the observation constants once more in CSR (landmark-major) order: the landmark-major linearisation (land_lin_body) reads them coalesced
__global__ __launch_bounds__(256) void k_bs_csrcopy(int n_obs, const int* __restrict__ pt_obs, const int* __restrict__ o_pose, const int* __restrict__ o_point,
                                                    const double* __restrict__ o_u, const double* __restrict__ o_v, const double* __restrict__ o_ur, const double* __restrict__ o_w,
                                                    const int* __restrict__ pose_slot, int* c_pose, int* c_point, int* c_slot, double* c_u, double* c_v, double* c_ur, double* c_w)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_obs) return;
    const int k = pt_obs[s];
    const int p = o_pose[k];
    c_pose[s] = p; c_point[s] = o_point[k]; c_slot[s] = pose_slot[p]; c_u[s] = o_u[k]; c_v[s] = o_v[k]; c_ur[s] = o_ur[k]; c_w[s] = o_w[k];
}

// pair (a, c), a <= c, of pose-block pair `blk` (row-major upper triangle of n_free x n_free)
__device__ __forceinline__ void bs_block_pair(int blk, int n_free, int* a, int* c)
{
    int pidx = blk, i = 0, rowlen = n_free;
    while (pidx >= rowlen) { pidx -= rowlen; --rowlen; ++i; }
    *a = i; *c = i + pidx;
}

// terms per pose-block pair: one wavefront per pair sums A[keyframe c][landmark] over the observations of keyframe a
__global__ __launch_bounds__(256) void k_bs_paircount(const int* __restrict__ A, int n_points, int n_free, int n_blocks, const int* __restrict__ free_pose,
                                                      const int* __restrict__ ps_start, const int* __restrict__ o_point, int* blk_count)
{
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= n_blocks) return;
    int a, c;
    bs_block_pair(blk, n_free, &a, &c);
    const int pa = free_pose[a], pc = free_pose[c];
    const int* Ac = A + (size_t)pc * n_points;
    int n = 0;
    for (int s = ps_start[pa] + lane; s < ps_start[pa + 1]; s += 64) n += Ac[o_point[s]] & 0xFFFF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) blk_count[blk] = n;
}

// blk_start = exclusive scan of blk_count (n_blocks + 1 entries), tickets cleared
__global__ __launch_bounds__(BS_THREADS) void k_bs_blkscan(const int* blk_count, int* blk_start, int* blk_ticket, int n_blocks)
{
    int carry = 0;
    for (int i0 = 0; i0 < n_blocks; i0 += BS_THREADS) {
        const int i = i0 + (int)threadIdx.x;
        const int x = i < n_blocks ? blk_count[i] : 0;
        int tot;
        const int pre = carry + bs_block_scan(x, &tot);
        if (i < n_blocks) { blk_start[i] = pre; blk_ticket[i] = 0; }
        carry += tot;
    }
    if (threadIdx.x == 0) blk_start[n_blocks] = carry;
}

// the pair lists: keyframe a's observations in storage (= landmark) order, each with its partner(s) in keyframe c
__global__ __launch_bounds__(256) void k_bs_pairfill(const int* __restrict__ A, const int* __restrict__ R, int n_points, int n_free, int n_blocks,
                                                     const int* __restrict__ free_pose, const int* __restrict__ ps_start, const int* __restrict__ o_point,
                                                     const int* __restrict__ blk_start, int4* blk_terms)
{
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= n_blocks) return;
    int a, c;
    bs_block_pair(blk, n_free, &a, &c);
    const int pa = free_pose[a], pc = free_pose[c];
    const int* Ac = A + (size_t)pc * n_points;
    const int* Rc = R + (size_t)pc * n_points;
    const int base_c = ps_start[pc];
    int out = blk_start[blk];
    const int s_end = ps_start[pa + 1];
    for (int s0 = ps_start[pa]; s0 < s_end; s0 += 64) {
        const int s = s0 + lane;
        int n = 0, j = 0, first = 0;
        if (s < s_end) { j = o_point[s]; n = Ac[j] & 0xFFFF; first = base_c + Rc[j]; }
        int incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        const int total = __shfl(incl, 63);
        int w = out + incl - n;
        for (int d = 0; d < n; ++d) blk_terms[w++] = make_int4(s, first + d, j, 0);
        out += total;
    }
}

// fills of the problem's zero-initialised block that are not zero: the identity rows of the reduced system below the rhs row
__global__ __launch_bounds__(256) void k_bs_identity(double* S, int dim, int dim_pad)
{
    const int r = dim + 1 + blockIdx.x * 256 + threadIdx.x;
    if (r < dim_pad) S[(size_t)r * dim_pad + r] = 1.0;
}

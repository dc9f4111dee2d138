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

// ---- everything the structure kernels of ONE problem need.  The kernels take an array of these and a problem index in blockIdx.y:
//      the windows of sixteen sessions are built by the same ~16 launches as one window (lpslam_hip_ba_build_batch), where a launch
//      chain per window was ~320 runtime calls per round from the sessions' set-up threads.  Launch extents are
//      the maxima over the batch; a workgroup beyond its problem's extent leaves at once.
struct BuildDesc {
    int n_poses, n_points, n_obs, n_free, n_blocks, dim, dim_pad, n_ord;
    int extra_cap, pad_;                                   // entries the table of further Schur parts has room for (behind the tickets: [count | items])
    const lpslam_hip_ba_obs* obs;
    int *A, *R, *pt_count, *ps_count, *ps_start, *pt_start, *slot_of, *pt_obs, *o_orig, *o_pose, *o_point;
    double *o_u, *o_v, *o_ur, *o_w;
    uint8_t *o_active, *act_in;
    const int *pose_slot, *free_pose;
    int *c_pose, *c_point, *c_slot;
    double *c_u, *c_v, *c_ur, *c_w;
    int *blk_count, *blk_start, *blk_ticket;
    int4* blk_terms;
    const int *band_order, *band_qinfo, *band_bstart;      // n_ord > 0: the window takes the band path (ba_band.inl)
    int4* band_ent;
    double* S;
    uint4* copy_dst; const uint4* copy_src; size_t copy_n16;      // inputs: page-locked staging -> the problem's block
    uint4* zero_dst; size_t zero_n16;                               // the zero-initialised part of the block
    const BaView* view;
};
constexpr int BUILD_MAX_BATCH = 64;         // descriptors a problem's block has room for (larger batches are built in chunks)

// the descriptors themselves: page-locked host memory -> device (one workgroup; the first launch of a build)
__global__ __launch_bounds__(256) void k_bs_descs_in(uint4* __restrict__ dst, const uint4* __restrict__ src, int n16)
{
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
}
// inputs over PCIe by load / store (not by the DMA engine: a copy packet queues behind whatever the engine is busy with -- the front
// end's image uploads held the next window's build back until the running solve had finished) and the zero fill, one launch each
__global__ __launch_bounds__(256) void k_bs_copy_in(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < d.copy_n16; i += (size_t)gridDim.x * 256) d.copy_dst[i] = d.copy_src[i];
}
__global__ __launch_bounds__(256) void k_bs_zero(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < d.zero_n16; i += (size_t)gridDim.x * 256) d.zero_dst[i] = z;
}

// A[p][j] += 1 per observation; landmark degrees; the identity rows of the reduced system below the rhs row
__global__ __launch_bounds__(256) void k_bs_count(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int k = blockIdx.x * 256 + threadIdx.x;
    { const int r = d.dim + 1 + k; if (r < d.dim_pad) d.S[(size_t)r * d.dim_pad + r] = 1.0; }
    if (k >= d.n_obs) return;
    const int p = d.obs[k].pose, j = d.obs[k].point;
    atomicAdd(&d.A[(size_t)p * d.n_points + j], 1);
    atomicAdd(&d.pt_count[j], 1);
}

// R[p][.] = exclusive scan of A[p][.] (one workgroup per keyframe); ps_count[p] = observations of keyframe p
__global__ __launch_bounds__(BS_THREADS) void k_bs_rowscan(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int p = blockIdx.x, n_points = d.n_points;
    if (p >= d.n_poses) return;
    const int* a = d.A + (size_t)p * n_points;
    int* r = d.R + (size_t)p * n_points;
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
    if (threadIdx.x == 0) d.ps_count[p] = carry;
}

// workgroup 0: ps_start = exclusive scan of ps_count (n_poses + 1 entries); workgroup 1: pt_start from pt_count
__global__ __launch_bounds__(BS_THREADS) void k_bs_starts(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int* in = blockIdx.x == 0 ? d.ps_count : d.pt_count;
    int* out = blockIdx.x == 0 ? d.ps_start : d.pt_start;
    const int n = blockIdx.x == 0 ? d.n_poses : d.n_points;
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
__global__ __launch_bounds__(256) void k_bs_scatter(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d.n_obs) return;
    const int p = d.obs[k].pose, j = d.obs[k].point;
    const size_t e = (size_t)p * d.n_points + j;
    const int base = d.ps_start[p] + d.R[e];
    const int c = d.A[e] & 0xFFFF;
    const int dd = c == 1 ? 0 : (atomicAdd(&d.A[e], 0x10000) >> 16);
    d.slot_of[base + dd] = k;
}

// second pass: storage position s takes the (s - first)-th smallest caller index of its group and copies that observation into
// the SoA arrays
__global__ __launch_bounds__(256) void k_bs_gather(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= d.n_obs) return;
    const int* slot_of = d.slot_of;
    int k = slot_of[s];
    const int p = d.obs[k].pose, j = d.obs[k].point;
    const size_t e = (size_t)p * d.n_points + j;
    const int c = d.A[e] & 0xFFFF;
    if (c > 1) {
        const int first = d.ps_start[p] + d.R[e], want = s - first;
        // rank selection among the c caller indices of the group (c is tiny; duplicates of one landmark in one keyframe are rare)
        for (int a = 0; a < c; ++a) {
            const int ka = slot_of[first + a];
            int rank = 0;
            for (int b2 = 0; b2 < c; ++b2) rank += slot_of[first + b2] < ka ? 1 : 0;
            if (rank == want) k = ka;
        }
    }
    const lpslam_hip_ba_obs o = d.obs[k];
    d.o_orig[s] = k; d.o_pose[s] = o.pose; d.o_point[s] = o.point;
    d.o_u[s] = o.u; d.o_v[s] = o.v; d.o_ur[s] = o.ur; d.o_w[s] = o.inv_sigma2;
    d.o_active[s] = 1; d.act_in[s] = 1;
}

// CSR by landmark: the keyframes of a landmark in order (column of A / R).  FOUR lanes per landmark, sixteen keyframes each: their
// counts and positions are loaded together (one round trip for 64 keyframes of a landmark), the lanes' totals meet by two DPP-row
// exchanges, every lane places its own keyframes' entries.  (One thread per landmark walking all keyframes was a dependent load pair
// per keyframe: 35 us for a 50-keyframe window on 80 wavefronts; sixteen loads in flight per thread: 25 us; this form: see DESIGN 13.3.)
__global__ __launch_bounds__(256) void k_bs_ptfill(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int gt = blockIdx.x * 256 + threadIdx.x;
    const int j = gt >> 2, part = gt & 3;
    const bool live = j < d.n_points;
    const int* __restrict__ A = d.A; const int* __restrict__ R = d.R; const int* __restrict__ ps_start = d.ps_start;
    int* __restrict__ pt_obs = d.pt_obs;
    const int n_poses = d.n_poses, n_points = d.n_points;
    int t = live ? d.pt_start[j] : 0;
    constexpr int U = 16;
    for (int p0 = 0; p0 < n_poses; p0 += 4 * U) {          // (uniform trip count: the exchanges below want all four lanes of a landmark)
        int c[U], first[U], mine = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = p0 + U * part + u, p = min(pp, n_poses - 1);
            const size_t e = (size_t)p * n_points + (live ? j : 0);
            c[u] = (live && pp < n_poses) ? (A[e] & 0xFFFF) : 0;
            first[u] = ps_start[p] + R[e];
            mine += c[u];
        }
        // exclusive prefix over the four lanes of the landmark (neighbours inside a quad)
        const int s1 = __shfl_xor(mine, 1), s2a = mine + s1;
        const int s2 = __shfl_xor(s2a, 2);
        const int before = (part & 1 ? s1 : 0) + (part & 2 ? s2 : 0);
        int w = t + before;
#pragma unroll
        for (int u = 0; u < U; ++u)
            for (int dd = 0; dd < c[u]; ++dd) pt_obs[w++] = first[u] + dd;
        t += s2a + s2;                                     // the four lanes' total
    }
}

// the observation constants once more in CSR (landmark-major) order: the landmark-major passes (k_ba_update, land_lin_body) read them coalesced
__global__ __launch_bounds__(256) void k_bs_csrcopy(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= d.n_obs) return;
    const int k = d.pt_obs[s];
    const int p = d.o_pose[k];
    d.c_pose[s] = p; d.c_point[s] = d.o_point[k]; d.c_slot[s] = d.pose_slot[p]; d.c_u[s] = d.o_u[k]; d.c_v[s] = d.o_v[k]; d.c_ur[s] = d.o_ur[k]; d.c_w[s] = d.o_w[k];
}

// pair (a, c), a <= c, of pose-block pair `blk` (row-major upper triangle of n_free x n_free)
__device__ __forceinline__ void bs_block_pair(int blk, int n_free, int* a, int* c)
{
    int pidx = blk, i = 0, rowlen = n_free;
    while (pidx >= rowlen) { pidx -= rowlen; --rowlen; ++i; }
    *a = i; *c = i + pidx;
}

// terms per pose-block pair: one wavefront per pair sums A[keyframe c][landmark] over the observations of keyframe a
__global__ __launch_bounds__(256) void k_bs_paircount(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= d.n_blocks) return;
    int a, c;
    bs_block_pair(blk, d.n_free, &a, &c);
    const int pa = d.free_pose[a], pc = d.free_pose[c];
    const int* Ac = d.A + (size_t)pc * d.n_points;
    int n = 0;
    for (int s = d.ps_start[pa] + lane; s < d.ps_start[pa + 1]; s += 64) n += Ac[d.o_point[s]] & 0xFFFF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) d.blk_count[blk] = n;
}

// blk_start = exclusive scan of blk_count (n_blocks + 1 entries), tickets cleared; behind the tickets the table of the lists' FURTHER
// parts (count, then block * SCH_MAXP + part in block order): k_ba_schur launches part 0 of every block and as many workgroups as this table
// can hold at most (terms / SCH_PART, known on the host) instead of three surplus workgroups per block that leave at once -- 3675 of them
// for the 1225 pairs of a 50-keyframe window, and dispatching them took longer than the work (the last workgroups started 12 us in).
__global__ __launch_bounds__(BS_THREADS) void k_bs_blkscan(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int n_blocks = d.n_blocks;
    int* extra = d.blk_ticket + n_blocks;
    int carry = 0, ecarry = 0;
    for (int i0 = 0; i0 < n_blocks; i0 += BS_THREADS) {
        const int i = i0 + (int)threadIdx.x;
        const int x = i < n_blocks ? d.blk_count[i] : 0;
        int tot, etot;
        const int pre = carry + bs_block_scan(x, &tot);
        const int more = schur_parts(x) - 1;
        const int epre = ecarry + bs_block_scan(more, &etot);
        if (i < n_blocks) {
            d.blk_start[i] = pre; d.blk_ticket[i] = 0;
            for (int q = 0; q < more; ++q) if (epre + q < d.extra_cap) extra[1 + epre + q] = SCH_MAXP * i + 1 + q;
        }
        carry += tot; ecarry += etot;
    }
    if (threadIdx.x == 0) { d.blk_start[n_blocks] = carry; if (n_blocks == 0) d.blk_start[1] = 0; extra[0] = ecarry < d.extra_cap ? ecarry : d.extra_cap; }
}

// the pair lists: keyframe a's observations in storage (= landmark) order, each with its partner(s) in keyframe c
__global__ __launch_bounds__(256) void k_bs_pairfill(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= d.n_blocks) return;
    int a, c;
    bs_block_pair(blk, d.n_free, &a, &c);
    const int pa = d.free_pose[a], pc = d.free_pose[c];
    const int* Ac = d.A + (size_t)pc * d.n_points;
    const int* Rc = d.R + (size_t)pc * d.n_points;
    const int base_c = d.ps_start[pc];
    int out = d.blk_start[blk];
    const int s_end = d.ps_start[pa + 1];
    for (int s0 = d.ps_start[pa]; s0 < s_end; s0 += 64) {
        const int s = s0 + lane;
        int n = 0, j = 0, first = 0;
        if (s < s_end) { j = d.o_point[s]; n = Ac[j] & 0xFFFF; first = base_c + Rc[j]; }
        int incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        const int total = __shfl(incl, 63);
        int w = out + incl - n;
        for (int dd = 0; dd < n; ++dd) d.blk_terms[w++] = make_int4(s, first + dd, j, 0);
        out += total;
    }
}

// the band path's entry table, in group order (ba_band.inl).  Thread per ordered landmark q: its observations (CSR order = keyframe
// order) become entries (storage slot, window row or -1 for a fixed keyframe, column | flags, landmark).
// flags: bit 16 = first entry of its landmark (writes the landmark's rhs vector), bit 17 = duplicate (same landmark seen
// twice by one keyframe: summed by the thread of the run's first entry), bit 18 = a duplicate follows.
__global__ __launch_bounds__(256) void k_bs_band_entries(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= d.n_ord) return;
    const int j = d.band_order[q], info = d.band_qinfo[q], f0 = info >> 8, li = info & 255;
    const int o0 = d.pt_start[j], o1 = d.pt_start[j + 1];
    int4* out = d.band_ent + d.band_bstart[q];
    int prev_pose = -1;
    for (int o = o0; o < o1; ++o) {
        const int s = d.pt_obs[o], p = d.o_pose[s], slot = d.pose_slot[p];
        int flags = 0;
        if (o == o0) flags |= 1 << 16;
        if (p == prev_pose) flags |= 1 << 17;
        if (o + 1 < o1 && d.o_pose[d.pt_obs[o + 1]] == p) flags |= 1 << 18;      // the next entry is a duplicate of this one
        prev_pose = p;
        out[o - o0] = make_int4(s, slot < 0 ? -1 : 6 * (slot - f0), 3 * li | flags, j);
    }
}

// state buffers <- the inputs, every observation active, control block cleared (k_ba_reset for every problem of the build)
__global__ __launch_bounds__(256) void k_bs_reset(const BuildDesc* __restrict__ descs)
{
    const BuildDesc& d = descs[blockIdx.y];
    const BaView& v = *d.view;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 7 * d.n_poses) v.poses_buf[0][i] = v.poses0[i];
    if (i < 3 * d.n_points) v.points_buf[0][i] = v.points0[i];
    if (i < d.n_obs) v.o_active[i] = 1;
    if (i == 0) {
        BaCtl c{};
        c.ni = 2; c.need_lin = 1; c.first = 1;
        *v.ctl = c;
    }
}

// the identity rows of the reduced system below the rhs row, on their own (lpslam_hip_ba_set_solver clears S)
__global__ __launch_bounds__(256) void k_bs_identity(double* S, int dim, int dim_pad)
{
    const int r = dim + 1 + blockIdx.x * 256 + threadIdx.x;
    if (r < dim_pad) S[(size_t)r * dim_pad + r] = 1.0;
}

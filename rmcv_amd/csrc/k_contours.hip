// k_contours.hip -- cv::findContours(binary, RETR_EXTERNAL, CHAIN_APPROX_NONE), the last step of
// rm::extract_color (/root/reference/src/imgproc.cpp:71-72), on the bit plane K1 leaves in HBM.
//
// OpenCV's Suzuki-Abe scanner is a sequential raster scan that rewrites the image with labels while it
// traces; its RETR_EXTERNAL rule ("skip an outer-border start if the last labelled pixel met on this
// row is positive") makes the result depend on that label state.  Restated on bit planes:
//   F    foreground (closed binary)                       read-only
//   LAB  pixel has been visited by a border trace         (label != 0, 1)
//   NEG  ... and at least one visit passed the east neighbour as zero (label 2|-128 instead of 2)
// A run start (F[x]=1, F[x-1]=0) that is unlabelled is a border-start candidate; it is accepted iff
// the nearest labelled pixel to its left on the row is NEG or there is none.  Only run starts and
// labelled pixels matter, so a row is scanned 64 pixels per operation.
//
// k_contours_literal: one wavefront per frame.  Lane l owns row 64*band + l of the current band and
// keeps the first acceptable candidate of its row; the wave repeatedly takes the raster-first one,
// traces it (border following on a 3-row x 64-bit register window of F), ORs the labels into
// LAB/NEG, and re-scans only the rows the trace touched below the start.  Exact for every input.
//
// Output is kept in DISCOVERY order (cont_start/cont_len); findContours order is its reverse.
#include "rmcv_internal.h"

namespace rmcv {

__device__ __forceinline__ uint64_t ld_l2(const uint64_t* p)
{ // bypass the (non-coherent) vector L1: labels are written with L2 atomics by another lane
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 64-bit window of row y starting at pixel xb (xb multiple of 32, may be -32 .. ) of a padded plane
__device__ __forceinline__ uint64_t win_load(const uint32_t* plane32, int prow, int y, int xb)
{
    const uint32_t* p = plane32 + ((int64_t)(y + 1) * prow + 1) * 2 + (xb >> 5);
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

struct Tracer {
    const uint32_t* F32;
    uint64_t* LAB;
    uint64_t* NEG;
    int prow;
    uint64_t r0, r1, r2; // rows y-1, y, y+1 of the window
    int xb;              // window origin (pixel), multiple of 32
    int x, y;

    __device__ __forceinline__ void recentre()
    {
        xb = ((x - 24) >> 5) * 32;
        r0 = win_load(F32, prow, y - 1, xb);
        r1 = win_load(F32, prow, y, xb);
        r2 = win_load(F32, prow, y + 1, xb);
    }
    // neighbour mask: bit s set <=> neighbour in direction s is foreground
    // s: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE  (y grows downward)
    __device__ __forceinline__ uint32_t nbmask() const
    {
        const int sh = x - xb - 1;
        uint32_t u = (uint32_t)(r0 >> sh) & 7u, m = (uint32_t)(r1 >> sh) & 7u, d = (uint32_t)(r2 >> sh) & 7u;
        return ((m >> 2) & 1u) | (((u >> 2) & 1u) << 1) | (((u >> 1) & 1u) << 2) | ((u & 1u) << 3) | ((m & 1u) << 4) |
               ((d & 1u) << 5) | (((d >> 1) & 1u) << 6) | (((d >> 2) & 1u) << 7);
    }
    __device__ __forceinline__ void move(int s)
    {
        const int dx = (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0);
        const int dy = (s >= 1 && s <= 3) ? -1 : ((s >= 5) ? 1 : 0);
        x += dx;
        y += dy;
        const int bx = x - xb;
        if (bx < 1 || bx > 62) {
            recentre();
        } else if (dy > 0) {
            r0 = r1; r1 = r2; r2 = win_load(F32, prow, y + 1, xb);
        } else if (dy < 0) {
            r2 = r1; r1 = r0; r0 = win_load(F32, prow, y - 1, xb);
        }
    }
    __device__ __forceinline__ void label(bool right_exit)
    {
        const int64_t idx = (int64_t)(y + 1) * prow + 1 + (x >> 6);
        const uint64_t bit = 1ull << (x & 63);
        atomicOr((unsigned long long*)(LAB + idx), (unsigned long long)bit);
        if (right_exit) atomicOr((unsigned long long*)(NEG + idx), (unsigned long long)bit);
    }
};

// icvFetchContour (outer border, CHAIN_APPROX_NONE) from start (x0,y0).  Writes at most `room` points
// to out (keeps tracing and labelling beyond that), returns the number of points; *ymax = lowest row.
__device__ int trace_border(Tracer& t, int x0, int y0, rmcv_point* out, int room, int* ymax)
{
    t.x = x0;
    t.y = y0;
    t.recentre();
    int n = 0, ym = y0;
    uint32_t nb = t.nbmask();
    // clockwise search for the first neighbour, starting just past west: s = 3,2,1,0,7,6,5,(4)
    int s = 4;
    do { s = (s - 1) & 7; } while (!((nb >> s) & 1u) && s != 4);
    if (s == 4) { // single pixel (west is background by construction)
        t.label(true);
        if (n < room) { out[n].x = x0; out[n].y = y0; }
        *ymax = y0;
        return 1;
    }
    const int dx1 = (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0);
    const int dy1 = (s >= 1 && s <= 3) ? -1 : ((s >= 5) ? 1 : 0);
    const int x1 = x0 + dx1, y1 = y0 + dy1; // i1
    for (;;) {
        const int s_end = s;
        const int k = (s_end + 1) & 7;
        const uint32_t rot = ((nb | (nb << 8)) >> k) & 0xFFu;
        s = (k + (__ffs((int)rot) - 1)) & 7; // counter-clockwise sweep from s_end+1 to the first foreground neighbour
        t.label((unsigned)(s - 1) < (unsigned)s_end);
        if (n < room) { out[n].x = t.x; out[n].y = t.y; }
        n++;
        const int cx = t.x, cy = t.y;
        t.move(s);
        if (t.y > ym) ym = t.y;
        if (t.x == x0 && t.y == y0 && cx == x1 && cy == y1) break; // i4 == i0 && i3 == i1
        nb = t.nbmask();
        s = (s + 4) & 7;
    }
    *ymax = ym;
    return n;
}

// first acceptable border-start candidate of row y at x >= xmin, or -1
__device__ int scan_row(const uint64_t* F, const uint64_t* LAB, const uint64_t* NEG, int prow, int ww, int y, int xmin)
{
    const int64_t base = (int64_t)(y + 1) * prow + 1;
    uint64_t carry = 0;
    bool last_pos = false; // no labelled pixel yet -> lnbd is the zero frame column -> accept
    for (int k = 0; k < ww; k++) {
        const uint64_t f = F[base + k];
        if (f == 0) { carry = 0; continue; }
        const uint64_t l = ld_l2(LAB + base + k), ng = ld_l2(NEG + base + k);
        uint64_t cand = f & ~((f << 1) | carry) & ~l;
        if (k * 64 + 63 < xmin) cand = 0;
        else if (k * 64 < xmin) cand &= ~0ull << (xmin - k * 64);
        while (cand) {
            const int b = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const uint64_t below = l & ((1ull << b) - 1);
            const bool pos = below ? !((ng >> (63 - __clzll((long long)below))) & 1ull) : last_pos;
            if (!pos) return k * 64 + b;
        }
        if (l) last_pos = !((ng >> (63 - __clzll((long long)l))) & 1ull);
        carry = f >> 63;
    }
    return -1;
}

__global__ __launch_bounds__(64) void k_contours_literal(const uint64_t* __restrict__ bits, uint64_t* lab, uint64_t* neg, int w,
                                                        int h, int ww, int prow, int64_t plane_pitch, rmcv_point* points,
                                                        int32_t* cont_start, int32_t* cont_len, int32_t* n_contours,
                                                        int32_t* n_points, int32_t* status, int max_contours, int max_points)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const uint64_t* F = bits + (int64_t)f * plane_pitch;
    uint64_t* LAB = lab + (int64_t)f * plane_pitch;
    uint64_t* NEG = neg + (int64_t)f * plane_pitch;
    rmcv_point* pts = points + (int64_t)f * max_points;
    int32_t* cs = cont_start + (int64_t)f * max_contours;
    int32_t* cl = cont_len + (int64_t)f * max_contours;
    Tracer t;
    t.F32 = reinterpret_cast<const uint32_t*>(F);
    t.LAB = LAB;
    t.NEG = NEG;
    t.prow = prow;

    int nc = 0, np = 0, st = 0; // wave-uniform
    for (int band = 0; band * 64 < h; band++) {
        const int y = band * 64 + lane;
        bool done = y >= h, dirty = true;
        int xmin = 0, found = -1;
        for (;;) {
            if (dirty && !done) {
                found = scan_row(F, LAB, NEG, prow, ww, y, xmin);
                dirty = false;
            }
            const uint64_t m = __ballot(!done && found >= 0);
            if (!m) break;
            const int L = __ffsll((long long)m) - 1;
            const int x0 = __shfl(found, L), y0 = band * 64 + L;
            if (lane < L) done = true; // rows above the start are behind the raster scan
            int len = 0, ymax = y0;
            if (lane == 0) {
                const int room = (nc < max_contours && np < max_points) ? (max_points - np) : 0;
                len = trace_border(t, x0, y0, pts + np, room, &ymax);
                if (nc < max_contours) { cs[nc] = np; cl[nc] = len; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // label atomics performed before the re-scan
            }
            len = __shfl(len, 0);
            ymax = __shfl(ymax, 0);
            if (nc >= max_contours) st |= RMCV_FRAME_OVF_CONTOURS;
            if (np + len > max_points) { st |= RMCV_FRAME_OVF_POINTS; }
            nc++;
            np = (np + len > max_points) ? max_points : np + len;
            if (lane == L) { xmin = x0 + 1; dirty = true; }
            else if (lane > L && y <= ymax) dirty = true;
        }
    }
    if (lane == 0) {
        n_contours[f] = nc < max_contours ? nc : max_contours;
        n_points[f] = np;
        status[f] = st;
    }
}

hipError_t launch_contours(const Geom& g, const Bufs& b, const Limits& lim, hipStream_t s)
{
    hipError_t e;
    const size_t plane_bytes = (size_t)g.n_frames * g.plane_pitch * sizeof(uint64_t);
    if ((e = hipMemsetAsync(b.lab, 0, plane_bytes, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(b.neg, 0, plane_bytes, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_contours_literal, dim3(g.n_frames), dim3(64), 0, s, b.bits, b.lab, b.neg, g.w, g.h, g.ww, g.prow,
                       g.plane_pitch, b.points, b.cont_start, b.cont_len, b.n_contours, b.n_points, b.status, lim.max_contours,
                       lim.max_points);
    return hipGetLastError();
}

// contours of every frame as CSR in findContours order (reverse discovery), for download
__global__ void k_pack_contours(const rmcv_point* __restrict__ points, const int32_t* __restrict__ cont_start,
                                const int32_t* __restrict__ cont_len, const int32_t* __restrict__ n_contours, int max_contours,
                                int max_points, rmcv_point* __restrict__ pts_out, int32_t* __restrict__ offs_out)
{
    const int f = blockIdx.x;
    const int n = n_contours[f];
    const int32_t* cs = cont_start + (int64_t)f * max_contours;
    const int32_t* cl = cont_len + (int64_t)f * max_contours;
    int32_t* offs = offs_out + (int64_t)f * (max_contours + 1);
    __shared__ int s_off;
    if (threadIdx.x == 0) {
        int o = 0;
        for (int i = 0; i < n; i++) {
            offs[i] = o;
            int len = cl[n - 1 - i];
            int room = max_points - cs[n - 1 - i];
            o += len < room ? len : (room > 0 ? room : 0);
        }
        offs[n] = o;
        s_off = o;
    }
    __syncthreads();
    for (int i = 0; i < n; i++) {
        const int k = n - 1 - i;
        const int o = offs[i], len = offs[i + 1] - offs[i];
        const rmcv_point* src = points + (int64_t)f * max_points + cs[k];
        rmcv_point* dst = pts_out + (int64_t)f * max_points + o;
        for (int j = threadIdx.x; j < len; j += blockDim.x) dst[j] = src[j];
    }
}

hipError_t launch_pack_contours(const Geom& g, const Bufs& b, const Limits& lim, rmcv_point* d_pts_out, int32_t* d_offs_out,
                                hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_contours, dim3(g.n_frames), dim3(256), 0, s, b.points, b.cont_start, b.cont_len, b.n_contours,
                       lim.max_contours, lim.max_points, d_pts_out, d_offs_out);
    return hipGetLastError();
}

} // namespace rmcv

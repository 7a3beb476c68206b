// Device side of SURVEY §8(f-3): the two array operations of SegDetectorRepresenter (src/postprocess.py) that sit
// between the GPU forward and the (unchanged, host) OpenCV contour code.
//   binarize_u8      postprocess.py:51-52    pred[:, 0] > thresh, as a uint8 bitmap (1 B/px over PCIe instead of 4)
//   box_scores       postprocess.py:186-198  box_score_fast for K boxes/polygons at once: mean of the probability map
//                                            over the cv2.fillPoly mask of each box (one workgroup per box)
// The fillPoly mask is evaluated per pixel in closed form: a pixel is set if it lies on the 8-connected Bresenham
// line of an edge (LineIterator, left-to-right) or inside the even-odd scanline fill in 16.16 fixed point
// (OpenCV drawing.cpp: fillPoly -> CollectPolyEdges + FillEdgeCollection, Line -> LineIterator; XY_SHIFT = 16).
#include "common.h"

namespace {

constexpr int MAX_PTS = 64;

__global__ void binarize_u8_kernel(const float* __restrict__ pred, long plane_stride, long hw, float thresh, int n_img,
                                   unsigned char* __restrict__ out) {
    const long total4 = (long)n_img * hw / 4;  // hw % 4 == 0
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long e = 4 * i, n = e / hw, r = e - n * hw;
        const f32x4 v = *reinterpret_cast<const f32x4*>(pred + n * plane_stride + r);
        uchar4 b;
        b.x = v[0] > thresh; b.y = v[1] > thresh; b.z = v[2] > thresh; b.w = v[3] > thresh;
        reinterpret_cast<uchar4*>(out)[i] = b;
    }
}

__device__ __forceinline__ long long trunc_div(long long a, long long b) { return a / b; }  // C++: toward zero, like OpenCV

// pixel (px,py) on the 8-connected line from (xa,ya) to (xb,yb)?
__device__ __forceinline__ bool on_line(int px, int py, int xa, int ya, int xb, int yb) {
    int dx = xb - xa, dy = yb - ya, x1 = xa, y1 = ya;
    if (dx < 0) { x1 = xb; y1 = yb; dx = -dx; dy = -dy; }
    const int sy = dy < 0 ? -1 : 1;
    dy = dy < 0 ? -dy : dy;
    if (dy > dx) {  // steep: one pixel per row
        const int i = (py - y1) * sy;
        if (i < 0 || i > dy) return false;
        const int m = (2 * dx * i + dy - 1) / (2 * dy);
        return px == x1 + m;
    }
    const int i = px - x1;
    if (i < 0 || i > dx) return false;
    const int m = dx == 0 ? 0 : (2 * dy * i + dx - 1) / (2 * dx);
    return py == y1 + sy * m;
}

__global__ __launch_bounds__(256) void box_score_kernel(const float* __restrict__ bitmap, int H, int W,
                                                        const float* __restrict__ boxes, int P, float* __restrict__ scores) {
    __shared__ int vx[MAX_PTS], vy[MAX_PTS];
    __shared__ int bb[4];
    __shared__ double rs[4], rc[4];
    const float* box = boxes + (long)blockIdx.x * P * 2;
    if (threadIdx.x == 0) {
        float x0 = box[0], x1 = box[0], y0 = box[1], y1 = box[1];
        for (int i = 1; i < P; ++i) {
            x0 = fminf(x0, box[2 * i]); x1 = fmaxf(x1, box[2 * i]);
            y0 = fminf(y0, box[2 * i + 1]); y1 = fmaxf(y1, box[2 * i + 1]);
        }
        bb[0] = min(max((int)floorf(x0), 0), W - 1);
        bb[1] = min(max((int)ceilf(x1), 0), W - 1);
        bb[2] = min(max((int)floorf(y0), 0), H - 1);
        bb[3] = min(max((int)ceilf(y1), 0), H - 1);
    }
    __syncthreads();
    const int xmin = bb[0], xmax = bb[1], ymin = bb[2], ymax = bb[3];
    if (threadIdx.x < P) {  // (box - min).astype(int32): truncation toward zero
        vx[threadIdx.x] = (int)(box[2 * threadIdx.x] - (float)xmin);
        vy[threadIdx.x] = (int)(box[2 * threadIdx.x + 1] - (float)ymin);
    }
    __syncthreads();
    const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
    double sum = 0.0, cnt = 0.0;
    for (int idx = threadIdx.x; idx < bw * bh; idx += blockDim.x) {
        const int py = idx / bw, px = idx - py * bw;
        bool in = false;
        int A = 0, B = 0;
        for (int i = 0; i < P; ++i) {
            const int j = i == 0 ? P - 1 : i - 1;
            int xa = vx[j], ya = vy[j], xb = vx[i], yb = vy[i];
            in = in || on_line(px, py, xa, ya, xb, yb);
            if (ya == yb) continue;
            if (ya > yb) { int t = xa; xa = xb; xb = t; t = ya; ya = yb; yb = t; }
            if (py < ya || py >= yb) continue;
            const long long dxf = trunc_div((long long)(xb - xa) << 16, (long long)(yb - ya));
            const long long xe = ((long long)xa << 16) + (long long)(py - ya) * dxf;
            A += ((xe + 65535) >> 16) <= px;
            B += (xe >> 16) < px;
        }
        in = in || A > B || (B & 1);
        if (in) {
            sum += (double)bitmap[(long)(ymin + py) * W + xmin + px];
            cnt += 1.0;
        }
    }
    sum = dbn_wave_sum_d(sum);
    cnt = dbn_wave_sum_d(cnt);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = sum; rc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = rs[0] + rs[1] + rs[2] + rs[3], c = rc[0] + rc[1] + rc[2] + rc[3];
        scores[blockIdx.x] = c > 0.0 ? (float)(s / c) : 0.f;  // cv2.mean over an empty mask is 0
    }
}

}  // namespace

extern "C" {

// out[n][h*w] = pred[n][0][h][w] > thresh (pred: NCHW with `channels` planes of h*w floats, h*w % 4 == 0)
int dbn_binarize_u8(const float* pred, int N, int channels, int H, int W, float thresh, unsigned char* out, void* stream) {
    DBN_REQUIRE(pred && out && N > 0 && channels > 0 && H > 0 && W > 0 && ((long)H * W) % 4 == 0);
    const long hw = (long)H * W;
    hipLaunchKernelGGL(binarize_u8_kernel, dim3(dbn_grid((long)N * hw / 4)), dim3(256), 0, (hipStream_t)stream, pred, channels * hw,
                       hw, thresh, N, out);
    return dbn_status();
}

// scores[k] = box_score_fast(bitmap[H][W], boxes[k][P][2] (x, y)), P <= 64 vertices per box (pad shorter polygons by
// repeating the last vertex: degenerate edges add nothing)
int dbn_box_scores(const float* bitmap, int H, int W, const float* boxes, int K, int P, float* scores, void* stream) {
    DBN_REQUIRE(bitmap && boxes && scores && H > 0 && W > 0 && K >= 0 && P >= 1 && P <= MAX_PTS);
    if (K == 0) return DBN_OK;
    hipLaunchKernelGGL(box_score_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, bitmap, H, W, boxes, P, scores);
    return dbn_status();
}

}  // extern "C"

// Micro-benchmark: marching-wave access pattern. Each wave reads 256 contiguous bytes (one dword per lane) of `streams`
// rows per step from a W x H byte image and waits for them `dist` steps later; `work` dependent VALU ops per step.
// Reports time per step and the implied load latency.   hipcc --offload-arch=gfx950 -O3 -o rowstream rowstream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int STREAMS, int DIST>
__global__ __launch_bounds__(256) void k(const uint8_t *img, int W, int H, int rows, int nstrips, int work, unsigned *out)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rb = wave / nstrips, strip = wave - rb * nstrips;
    const int c0 = strip * 232 + 4 * lane;
    unsigned acc = 0;
    uint32_t q[STREAMS][DIST];
    const int y0 = rb * rows;
    for (int f = 0; f < DIST; f++)
        for (int s = 0; s < STREAMS; s++) q[s][f] = *(const uint32_t *)(img + (size_t)s * W * H + (size_t)min(y0 + f, H - 1) * W + c0);
    for (int y = y0; y < y0 + rows; y++) {
        uint32_t cur[STREAMS];
        for (int s = 0; s < STREAMS; s++) {
            cur[s] = q[s][0];
            for (int f = 0; f + 1 < DIST; f++) q[s][f] = q[s][f + 1];
            q[s][DIST - 1] = *(const uint32_t *)(img + (size_t)s * W * H + (size_t)min(y + DIST, H - 1) * W + c0);
        }
        unsigned v = 0;
        for (int s = 0; s < STREAMS; s++) v += cur[s];
        for (int i = 0; i < work; i++) v = v * 1664525u + 1013904223u;
        acc += v;
    }
    out[wave * 64 + lane] = acc;
}

template <int STREAMS, int DIST> void run(const uint8_t *img, int W, int H, int rows, int work, unsigned *out)
{
    const int nstrips = W / 232;
    const int nitems = nstrips * (H / rows);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<STREAMS, DIST><<<(nitems + 3) / 4, 256>>>(img, W, H, rows, nstrips, work, out);
    hipEventRecord(e0);
    k<STREAMS, DIST><<<(nitems + 3) / 4, 256>>>(img, W, H, rows, nstrips, work, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rounds = (double)nitems / 4096.0;
    printf("streams=%d dist=%d rows=%d work=%4d waves=%5d: %.3f ms  -> %.2f us per row-step (per resident wave), %.2f TB/s\n", STREAMS, DIST, rows, work,
           nitems, ms, ms * 1e3 / (rows * (rounds < 1 ? 1 : rounds)), (double)nitems * rows * 256.0 * STREAMS / (ms * 1e-3) / 1e12);
}

int main()
{
    const int W = 10980, H = 10980;
    uint8_t *img; unsigned *out;
    hipMalloc(&img, (size_t)3 * W * H + 4096); hipMemset(img, 1, (size_t)3 * W * H);
    hipMalloc(&out, 64 * 4 * 100000);
    for (int work : {0, 200, 500}) {
        run<1, 1>(img, W, H, 127, work, out);
        run<3, 1>(img, W, H, 127, work, out);
        run<3, 4>(img, W, H, 127, work, out);
        run<3, 1>(img, W, H, 32, work, out);
        run<3, 4>(img, W, H, 32, work, out);
    }
    return 0;
}

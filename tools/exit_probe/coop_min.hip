// coop_min.hip -- the smallest program that can show the exit-time fault of round 2 (gpurun_out/rocprof_segv.log):
// one kernel launched with hipLaunchCooperativeKernel (argv[1] = "coop") or as a plain launch ("plain"), everything the
// program created released again (argv[2] = "release") or left to the runtime's exit handlers ("leak"), then a normal
// return from main.  No librri_hip, no Python: whatever happens at exit() belongs to the HIP runtime and to the tool
// that wraps it.  /proc/self/maps goes to argv[3] just before main returns, so the frames of a fault can be resolved.
//   hipcc -O2 --offload-arch=gfx950 tools/exit_probe/coop_min.hip -o tools/exit_probe/coop_min
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void k_touch(unsigned* p) {
    if (threadIdx.x == 0) atomicAdd(p, 1u);
}

static void dump_maps(const char* path) {
    FILE* in = fopen("/proc/self/maps", "r");
    FILE* out = fopen(path, "w");
    if (!in || !out) return;
    char buf[4096];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, in)) > 0) fwrite(buf, 1, got, out);
    fclose(in);
    fclose(out);
}

int main(int argc, char** argv) {
    const bool coop = argc > 1 && !strcmp(argv[1], "coop");
    const bool release = argc > 2 && !strcmp(argv[2], "release");
    unsigned* d = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t ev = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 2;
    if (hipMalloc((void**)&d, 256) != hipSuccess) return 2;
    (void)hipMemsetAsync(d, 0, 256, s);
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipError_t e;
    if (coop) {
        void* args[] = {(void*)&d};
        e = hipLaunchCooperativeKernel((const void*)k_touch, dim3(256), dim3(512), args, 0, s);
    } else {
        hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, s, d);
        e = hipGetLastError();
    }
    (void)hipEventRecord(ev, s);
    unsigned h = 0;
    (void)hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    printf("%s launch: %s, %u workgroups ran; %s\n", coop ? "cooperative" : "plain", hipGetErrorString(e), h,
           release ? "releasing stream, event, buffer" : "leaving stream, event, buffer to exit()");
    if (release) {
        (void)hipEventDestroy(ev);
        (void)hipFree(d);
        (void)hipStreamDestroy(s);
    }
    if (argc > 3) dump_maps(argv[3]);
    fflush(stdout);
    return 0;
}

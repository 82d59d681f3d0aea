#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <cstdio>
#include <string>
#include <vector>
static const char* src = R"(
namespace t {
template <int K, typename T>
__global__ __launch_bounds__(256) void big_lds(T* out, int n) {
  extern __shared__ __align__(16) unsigned char sm[];
  float* f = reinterpret_cast<float*>(sm);
  for (int i = threadIdx.x; i < n; i += blockDim.x) f[i] = (float)(i * K);
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += f[n - 1 - i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (T)s;
}
}
)";
#define CK(x) do { auto e = (x); if (e != 0) { printf("fail %s -> %d\n", #x, (int)e); return 1; } } while (0)
int main(int argc, char** argv) {
  hiprtcProgram prog;
  CK(hiprtcCreateProgram(&prog, src, "spike.hip", 0, nullptr, nullptr));
  const char* name = "t::big_lds<3, float>";
  CK(hiprtcAddNameExpression(prog, name));
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  auto rc = hiprtcCompileProgram(prog, 3, opts);
  size_t ls; hiprtcGetProgramLogSize(prog, &ls); std::string log(ls, 0); hiprtcGetProgramLog(prog, log.data());
  if (rc) { printf("compile failed: %s\n", log.c_str()); return 1; }
  const char* lowered; CK(hiprtcGetLoweredName(prog, name, &lowered));
  size_t cs; CK(hiprtcGetCodeSize(prog, &cs)); std::vector<char> code(cs); CK(hiprtcGetCode(prog, code.data()));
  printf("compiled: %zu bytes, lowered %s\n", cs, lowered);
  if (argc > 1) return 0;  // compile only (no GPU)
  hipModule_t mod; CK(hipModuleLoadData(&mod, code.data()));
  hipFunction_t fn; CK(hipModuleGetFunction(&fn, mod, lowered));
  const int n = 150 * 1024 / 4; const size_t lds = (size_t)n * 4;
  auto e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  printf("hipFuncSetAttribute on hipFunction_t -> %d (%s)\n", (int)e, hipGetErrorString(e)); (void)hipGetLastError();
  float* out; CK(hipMalloc(&out, 4 * 256 * 4));
  void* args[] = {&out, (void*)&n};
  e = hipModuleLaunchKernel(fn, 4, 1, 1, 256, 1, 1, lds, 0, args, nullptr);
  printf("launch -> %d (%s)\n", (int)e, hipGetErrorString(e));
  CK(hipDeviceSynchronize());
  float h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  printf("out[0] = %g (expect > 0)\n", h[0]);
  return 0;
}

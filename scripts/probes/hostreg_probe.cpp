// Which forms of hipHostRegister this machine accepts (malloc'd, page-aligned, mmap'd memory; flag variants).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
int main() {
  size_t n = 1 << 20;
  void* a = malloc(n + 4096);
  void* al = (void*)(((uintptr_t)a + 4095) & ~(uintptr_t)4095);
  void* mm = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  void* pm = nullptr; posix_memalign(&pm, 2 << 20, 4 << 20);
  unsigned flags[] = {hipHostRegisterDefault, hipHostRegisterPortable, hipHostRegisterMapped, hipHostRegisterPortable | hipHostRegisterMapped};
  struct { const char* name; void* p; size_t n; } cases[] = {{"malloc+1", (char*)a + 1, n}, {"aligned", al, n}, {"mmap", mm, n}, {"memalign2M", pm, 4u << 20}};
  hipSetDevice(0); hipFree(0);
  for (auto& c : cases)
    for (unsigned f : flags) {
      ((volatile char*)c.p)[0] = 1;
      hipError_t e = hipHostRegister(c.p, c.n, f);
      printf("%-10s flags %u -> %s\n", c.name, f, hipGetErrorString(e));
      if (e == hipSuccess) hipHostUnregister(c.p); else (void)hipGetLastError();
    }
  void* hp = nullptr; hipError_t e = hipHostMalloc(&hp, n, hipHostMallocDefault); printf("hipHostMalloc -> %s\n", hipGetErrorString(e));
  return 0;
}

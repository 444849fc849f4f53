#include <sys/mman.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
int main(){
  const size_t sz = size_t(1)<<30;
  for (int huge=0; huge<2; ++huge){
    void* p = mmap(nullptr, sz, PROT_READ|PROT_WRITE, MAP_PRIVATE|MAP_ANONYMOUS, -1, 0);
    if (huge) printf("madvise rc=%d\n", madvise(p, sz, MADV_HUGEPAGE));
    auto t0=std::chrono::steady_clock::now();
    memset(p, 1, sz);
    auto t1=std::chrono::steady_clock::now();
    memset(p, 2, sz);
    auto t2=std::chrono::steady_clock::now();
    printf("huge=%d first touch %.3f s, second %.3f s\n", huge, std::chrono::duration<double>(t1-t0).count(), std::chrono::duration<double>(t2-t1).count());
    munmap(p, sz);
  }
}

#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main(){ auto t0=std::chrono::steady_clock::now(); auto el=[&]{return std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();};
 void*p; hipHostMalloc(&p, 1<<20, 0); printf("first hipHostMalloc(1MB): %.3f s\n", el());
 void*q; hipHostMalloc(&q, 90<<20, 0); printf("hipHostMalloc(90MB): %.3f s\n", el());
 void*d; hipMalloc(&d, 1<<30); printf("hipMalloc(1GB): %.3f s\n", el());
 hipStream_t s; hipStreamCreate(&s); printf("stream: %.3f s\n", el());
 hipMemcpyAsync(d,q,90<<20,hipMemcpyHostToDevice,s); hipStreamSynchronize(s); printf("h2d 90MB: %.3f s\n", el());
 return 0; }

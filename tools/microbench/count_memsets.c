// LD_PRELOAD interposer: counts hipMemset* calls (the HIP runtime entry points a captured stream turns into memset nodes).
// build: gcc -O2 -shared -fPIC tools/microbench/count_memsets.c -o tools/bin/libcount_memsets.so -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
static long n_async, n_sync, n_d32, n_d8, n_2d;
long probe_memset_calls(void) { return n_async + n_sync + n_d32 + n_d8 + n_2d; }
#define NEXT(name) static int (*real)() = 0; if (!real) real = (int (*)())dlsym(RTLD_NEXT, name)
int hipMemsetAsync(void* p, int v, size_t n, void* s) { NEXT("hipMemsetAsync"); ++n_async; return real(p, v, n, s); }
int hipMemset(void* p, int v, size_t n) { NEXT("hipMemset"); ++n_sync; return real(p, v, n); }
int hipMemsetD32Async(void* p, int v, size_t n, void* s) { NEXT("hipMemsetD32Async"); ++n_d32; return real(p, v, n, s); }
int hipMemsetD8Async(void* p, unsigned char v, size_t n, void* s) { NEXT("hipMemsetD8Async"); ++n_d8; return real(p, v, n, s); }
int hipMemset2DAsync(void* p, size_t pitch, int v, size_t w, size_t h, void* s) { NEXT("hipMemset2DAsync"); ++n_2d; return real(p, pitch, v, w, h, s); }
__attribute__((destructor)) static void report(void) { fprintf(stderr, "[count_memsets] hipMemsetAsync %ld hipMemset %ld D32Async %ld D8Async %ld 2DAsync %ld\n", n_async, n_sync, n_d32, n_d8, n_2d); }

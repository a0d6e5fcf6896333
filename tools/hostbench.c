/* hostbench.c — H host threads calling a SAF-style process entry point (X_process(h, in, out, nIn, nOut, nSamples), host
 * pointers) on H handles at once, M calls each, natively: Python threads serialise on the interpreter lock between calls, which
 * hides what the library does.  Built on demand by bench.py:  gcc -O2 -shared -fPIC -pthread tools/hostbench.c -o <so> */
#include <pthread.h>
#include <time.h>

typedef void (*process_fn)(void*, const float* const*, float**, int, int, int);
struct job { process_fn fn; void* h; const float* const* in; float** out; int nIn, nOut, F, calls; pthread_barrier_t* bar; };

static void* worker(void* p)
{
    struct job* j = (struct job*)p;
    pthread_barrier_wait(j->bar);
    for (int i = 0; i < j->calls; i++) j->fn(j->h, j->in, j->out, j->nIn, j->nOut, j->F);
    return 0;
}

/* seconds from the common start to the last thread's end; < 0 on failure */
double hostbench_run(void* fn, void** handles, const float* const** ins, float*** outs, int H, int nIn, int nOut, int F, int calls)
{
    if (H < 1 || H > 256) return -1.0;
    pthread_t th[256];
    struct job jobs[256];
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, 0, (unsigned)H + 1);
    for (int i = 0; i < H; i++) {
        jobs[i].fn = (process_fn)fn; jobs[i].h = handles[i]; jobs[i].in = ins[i]; jobs[i].out = outs[i];
        jobs[i].nIn = nIn; jobs[i].nOut = nOut; jobs[i].F = F; jobs[i].calls = calls; jobs[i].bar = &bar;
        if (pthread_create(&th[i], 0, worker, &jobs[i])) return -1.0;
    }
    struct timespec t0, t1;
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < H; i++) pthread_join(th[i], 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    pthread_barrier_destroy(&bar);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

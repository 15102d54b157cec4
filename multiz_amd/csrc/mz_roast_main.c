/* mz_roast: the tree driver's command line on libmzamd.so (reference roast / auto_mz.c).
 * The library is loaded at run time, AFTER the OpenMP wait policy is set: the driver's host stages alternate between loops on all
 * threads and stretches of one thread, and libgomp's idle threads by default spin for a while before they sleep -- on a box that
 * gives the process 16 CPUs' worth of time for 32 threads that spinning is taken from the threads that work (2.8 s -> 2.5 s for the
 * 30-species run of tests/tools/roast_big.py).  libgomp reads OMP_WAIT_POLICY once, when it is loaded; a value the user set is kept. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int main(int argc, char **argv)
{
    void *lib;
    int (*run)(int, char **);
    void (*warm_wait)(void);
    int rc;
    setenv("OMP_WAIT_POLICY", "passive", 0);
    {   /* the library beside the program (by its path: a dlopen() that an interposed one makes -- a sanitizer's -- does not see the run path) */
        char path[4096 + 16];
        ssize_t n = readlink("/proc/self/exe", path, 4096);
        lib = NULL;
        if (n > 0) {
            while (n > 0 && path[n - 1] != '/') --n;
            strcpy(path + n, "libmzamd.so");
            lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!lib) lib = dlopen("libmzamd.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!lib) { fprintf(stderr, "mz_roast: %s\n", dlerror()); return 1; }
    *(void **)&run = dlsym(lib, "mz_roast_main");
    *(void **)&warm_wait = dlsym(lib, "mz_warm_wait");
    if (!run || !warm_wait) { fprintf(stderr, "mz_roast: %s\n", dlerror()); return 1; }
    rc = run(argc, argv);
    warm_wait();
    return rc;
}

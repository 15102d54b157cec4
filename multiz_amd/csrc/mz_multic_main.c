/* mz_multic: the multic command line on libmzamd.so (reference multic.c main(), :259-403) */
#include "../../include/mz_multiz.h"
int main(int argc, char **argv) { return mz_multic_main(argc, argv); }

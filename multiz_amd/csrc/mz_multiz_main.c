/* mz_multiz: the multiz command line on libmzamd.so (reference multiz.c main(), :180-294) */
#include "../../include/mz_multiz.h"
int main(int argc, char **argv) { return mz_multiz_main(argc, argv); }

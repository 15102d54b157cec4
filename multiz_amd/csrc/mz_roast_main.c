#include "../../include/mz_multiz.h"
void mz_warm_wait(void);      /* (include/mz_amd.h) */
int main(int argc, char **argv) { const int rc = mz_roast_main(argc, argv); mz_warm_wait(); return rc; }

#include "../../include/mz_multiz.h"
int main(int argc, char **argv) { return mz_roast_main(argc, argv); }

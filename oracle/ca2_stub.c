/* oracle/ca2_stub.c -- TEST INFRASTRUCTURE.  pre_yama2() asks connectionAgreement2() (reference align_util.c:520-)
 * whether the pairwise files support a merge; that function needs the whole pwuAliFiles machinery of tba.  For the
 * differential test of pre_yama2 itself (tests/test_preyama.py) this stand-in, loaded RTLD_GLOBAL ahead of both the
 * compiled reference and libmzamd.so, simply says yes. */
struct mafAli;
struct pwuAliFiles;
int connectionAgreement2(struct mafAli *a2, struct mafAli *a3, int cbeg2, int cend2, int cbeg3, int cend3, struct pwuAliFiles *pws)
{
    (void)a2; (void)a3; (void)cbeg2; (void)cend2; (void)cbeg3; (void)cend3; (void)pws;
    return 1;
}

/* oracle/ref_glue.c -- TEST INFRASTRUCTURE ONLY.  The reference's src/util.c, compiled as it lies into
 * oracle/_ref/libref_util.so, imports one global, BC_LEN: the platform's barcode length, which the reference's
 * src/main.c:30 defines and :325 sets from the platform profile.  main.c cannot be part of the build (it pulls in the
 * whole program and bwa), so the parameter lives here.  No algorithm in this file. */
int BC_LEN = 16;
void ref_set_bc_len(int n) { BC_LEN = n; }

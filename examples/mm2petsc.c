/* MatrixMarket -> PETSc binary, so that SuiteSparse matrices (e.g. BASELINE.json's Flan_1565) can be staged for
 * MatLoad (the reference's own converter is the example src/mat/examples/tests/ex72.c, symmetric input
 * only).  Host-only: no GPU is touched.
 *
 *   mm2petsc -fin A.mtx -fout A.petsc
 *
 * Handles `%%MatrixMarket matrix coordinate <real|integer|pattern> <general|symmetric|skew-symmetric>`; a symmetric
 * file's strictly lower (or upper) entries are mirrored, pattern entries get the value 1.  The file is read twice: once
 * to count the entries of every row (exact preallocation), once to insert them. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "petschipmi355x.h"
#include "petscmini_mpinames.h"   /* MPI_Comm spelling on the harness */

#define CHK(call) do { PetscErrorCode e_ = (call); if (e_) { fprintf(stderr, "error %d: %s\n", (int)e_, PetscGetLastErrorMessage()); return 1; } } while (0)

static void lower(char *s) { for (; *s; s++) *s = (char)tolower((unsigned char)*s); }

int main(int argc, char **argv) {
  const char *fin = NULL, *fout = NULL;
  for (int k = 1; k + 1 < argc; k++) {
    if (!strcmp(argv[k], "-fin")) fin = argv[++k];
    else if (!strcmp(argv[k], "-fout")) fout = argv[++k];
  }
  if (!fin || !fout) { fprintf(stderr, "usage: %s -fin <matrix.mtx> -fout <matrix.petsc>\n", argv[0]); return 2; }
  FILE *f = fopen(fin, "r");
  if (!f) { fprintf(stderr, "cannot open %s\n", fin); return 2; }
  char line[1024], obj[64] = "", fmt[64] = "", field[64] = "real", symm[64] = "general";
  if (!fgets(line, sizeof(line), f)) { fprintf(stderr, "%s: empty file\n", fin); return 2; }
  if (!strncmp(line, "%%MatrixMarket", 14)) {
    sscanf(line + 14, "%63s %63s %63s %63s", obj, fmt, field, symm);
    lower(obj); lower(fmt); lower(field); lower(symm);
    if (strcmp(obj, "matrix") || strcmp(fmt, "coordinate")) { fprintf(stderr, "%s: only 'matrix coordinate' files are supported\n", fin); return 2; }
    if (!strcmp(field, "complex")) { fprintf(stderr, "%s: complex matrices are not supported (real scalars)\n", fin); return 2; }
    if (!fgets(line, sizeof(line), f)) { fprintf(stderr, "%s: truncated\n", fin); return 2; }
  }
  while (line[0] == '%') if (!fgets(line, sizeof(line), f)) { fprintf(stderr, "%s: truncated\n", fin); return 2; }
  long M, N, NZ;
  if (sscanf(line, "%ld %ld %ld", &M, &N, &NZ) != 3 || M <= 0 || N <= 0 || NZ < 0 || M > 2147483647L || N > 2147483647L) { fprintf(stderr, "%s: bad size line\n", fin); return 2; }
  const int pattern = !strcmp(field, "pattern");
  const int mirror = !strcmp(symm, "symmetric") ? 1 : (!strcmp(symm, "skew-symmetric") ? -1 : 0);
  const long data_start = ftell(f);

  CHK(PetscHIPMI355XInitialize(-1));
  PetscInt *cnt = (PetscInt *)calloc((size_t)M, sizeof(PetscInt));
  if (!cnt) { fprintf(stderr, "out of memory\n"); return 1; }
  for (int pass = 0; pass < 2; pass++) {                 /* pass 0: count, pass 1: insert */
    static Mat A;
    if (pass == 1) {
      CHK(MatCreate(PETSC_COMM_SELF, &A));
      CHK(MatSetSizes(A, (PetscInt)M, (PetscInt)N, (PetscInt)M, (PetscInt)N));
      CHK(MatSetType(A, MATSEQAIJHIPMI355X));
      CHK(MatSeqAIJSetPreallocation(A, 0, cnt));
    }
    fseek(f, data_start, SEEK_SET);
    for (long e = 0; e < NZ; e++) {
      long i, j; double v = 1.0;
      int got = pattern ? fscanf(f, "%ld %ld", &i, &j) : fscanf(f, "%ld %ld %lf", &i, &j, &v);
      if (got != (pattern ? 2 : 3) || i < 1 || i > M || j < 1 || j > N) { fprintf(stderr, "%s: bad entry %ld\n", fin, e + 1); return 2; }
      PetscInt r = (PetscInt)(i - 1), c = (PetscInt)(j - 1);
      PetscScalar val = v;
      if (pass == 0) { cnt[r]++; if (mirror && r != c && c < M) cnt[c]++; }
      else {
        CHK(MatSetValues(A, 1, &r, 1, &c, &val, INSERT_VALUES));
        if (mirror && r != c) { PetscScalar w = mirror > 0 ? val : -val; CHK(MatSetValues(A, 1, &c, 1, &r, &w, INSERT_VALUES)); }
      }
    }
    if (pass == 1) {
      PetscViewer view;
      CHK(MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY));
      CHK(MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY));
      CHK(PetscViewerBinaryOpen(PETSC_COMM_SELF, fout, FILE_MODE_WRITE, &view));
      CHK(MatView(A, view));
      CHK(PetscViewerDestroy(&view));
      CHK(MatDestroy(&A));
    }
  }
  free(cnt);
  fclose(f);
  printf("%s: %ld x %ld, %ld stored entries (%s %s) -> %s\n", fin, M, N, NZ, field, symm, fout);
  CHK(PetscHIPMI355XFinalize());
  return 0;
}

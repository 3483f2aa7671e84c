/* PETSc binary format (big-endian on disk): MatLoad / MatView for AIJ and VecLoad / VecView, SURVEY 8f.2.
 * Layout (src/mat/impls/aij/seq/aij.c:4093-4157, include/petscmat.h:136, include/petscvec.h:113):
 *   Mat: int32 {MAT_FILE_CLASSID = 1211216, M, N, nz}, int32 rowlens[M], int32 cols[nz], float64 vals[nz]
 *   Vec: int32 {VEC_FILE_CLASSID = 1211214, n}, float64 vals[n]
 * Parallel MatLoad (MatLoad_MPIAIJ, mpiaij.c:3416): here every rank reads the header and the row lengths and then
 * seeks to its own rows (the reference has rank 0 read and send). */
#include "petscimpl.h"
#include <stdint.h>

#define MAT_FILE_CLASSID 1211216
#define VEC_FILE_CLASSID 1211214

struct _p_PetscViewer { PetscComm comm; FILE *f; int mode; };

static uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }
static uint64_t bswap64(uint64_t v) { return ((uint64_t)bswap32((uint32_t)v) << 32) | bswap32((uint32_t)(v >> 32)); }
static int little_endian(void) { const uint16_t one = 1; return *(const unsigned char *)&one; }

static PetscErrorCode read_ints(FILE *f, PetscInt *p, size_t n) {
  if (fread(p, sizeof(PetscInt), n, f) != n) SETERRQ(0, 66 /* PETSC_ERR_FILE_READ */, "Read past end of file");
  if (little_endian()) for (size_t k = 0; k < n; k++) p[k] = (PetscInt)bswap32((uint32_t)p[k]);
  return 0;
}
static PetscErrorCode read_scalars(FILE *f, PetscScalar *p, size_t n) {
  if (fread(p, sizeof(PetscScalar), n, f) != n) SETERRQ(0, 66, "Read past end of file");
  if (little_endian()) { uint64_t *q = (uint64_t *)p; for (size_t k = 0; k < n; k++) q[k] = bswap64(q[k]); }
  return 0;
}
static PetscErrorCode write_ints(FILE *f, const PetscInt *p, size_t n) {
  for (size_t k = 0; k < n; k++) { uint32_t v = little_endian() ? bswap32((uint32_t)p[k]) : (uint32_t)p[k]; if (fwrite(&v, 4, 1, f) != 1) SETERRQ(0, 67, "write failed"); }
  return 0;
}
static PetscErrorCode write_scalars(FILE *f, const PetscScalar *p, size_t n) {
  for (size_t k = 0; k < n; k++) { uint64_t v; memcpy(&v, &p[k], 8); if (little_endian()) v = bswap64(v); if (fwrite(&v, 8, 1, f) != 1) SETERRQ(0, 67, "write failed"); }
  return 0;
}

PetscErrorCode PetscViewerBinaryOpen(PetscComm comm, const char name[], PetscFileMode mode, PetscViewer *viewer) {
  PetscViewer v;
  PetscErrorCode ierr = PetscMalloc(sizeof(*v), &v);CHKERRQ(ierr);
  v->comm = comm; v->mode = (int)mode;
  v->f = fopen(name, mode == FILE_MODE_READ ? "rb" : "wb");
  if (!v->f) { free(v); SETERRQ(comm, 65 /* PETSC_ERR_FILE_OPEN */, "Cannot open file %s", name); }
  *viewer = v;
  return 0;
}
PetscErrorCode PetscViewerDestroy(PetscViewer *viewer) {
  if (*viewer) { if ((*viewer)->f) fclose((*viewer)->f); free(*viewer); *viewer = NULL; }
  return 0;
}

extern PetscErrorCode MatSeqAIJSetPreallocationCSR(Mat, const PetscInt[], const PetscInt[], const PetscScalar[]);
extern PetscErrorCode MatMPIAIJSetPreallocationCSR(Mat, const PetscInt[], const PetscInt[], const PetscScalar[]);

/* the validated part of MatLoad: everything that can fail after the first allocation, so that the caller frees once */
static PetscErrorCode matload_body(Mat A, PetscViewer viewer, long base, const PetscInt header[4], PetscInt *rowlens,
                                   PetscInt **li_, PetscInt **lj_, PetscScalar **la_) {
  PetscErrorCode ierr;
  const PetscInt M = header[1], N = header[2], nz = header[3];
  PetscInt *li, *lj; PetscScalar *la;
  ierr = read_ints(viewer->f, rowlens, (size_t)M);CHKERRQ(ierr);
  {   /* aij.c:4125: the row lengths must add up to the header's nonzero count (and none may be negative) */
    long sum = 0;
    for (PetscInt r = 0; r < M; r++) {
      if (rowlens[r] < 0 || rowlens[r] > N) SETERRQ(A->comm, 79, "Inconsistant matrix data in file: row %d has length %d", r, rowlens[r]);
      sum += rowlens[r];
    }
    if (sum != (long)nz) SETERRQ(A->comm, 66, "Inconsistant matrix data in file. no-nonzeros = %d, sum-row-lengths = %ld", nz, sum);
  }
  if (!A->type_name[0]) {
    if (A->m_req == -1 && A->M_req == -1) { ierr = MatSetSizes(A, PETSC_DECIDE, PETSC_DECIDE, M, N);CHKERRQ(ierr); }   /* applies a type chosen earlier */
    if (!A->type_name[0]) { ierr = MatSetType(A, MATAIJ);CHKERRQ(ierr); }
  }
  if (A->rmap->N != M || A->cmap->N != N) SETERRQ(A->comm, 79, "Matrix in file of different length (%d,%d) than the input matrix (%d,%d)", M, N, A->rmap->N, A->cmap->N);
  PetscInt rs = A->rmap->rstart, re = A->rmap->rend, m = re - rs;
  long before = 0, mine = 0;
  for (PetscInt r = 0; r < rs; r++) before += rowlens[r];
  for (PetscInt r = rs; r < re; r++) mine += rowlens[r];
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(m + 1), li_);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(mine, 1), lj_);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(mine, 1), la_);CHKERRQ(ierr);
  li = *li_; lj = *lj_; la = *la_;
  li[0] = 0;
  for (PetscInt r = 0; r < m; r++) li[r + 1] = li[r] + rowlens[rs + r];
  long cols0 = base + 16 + 4L * M, vals0 = cols0 + 4L * nz;
  if (fseek(viewer->f, cols0 + 4L * before, SEEK_SET)) SETERRQ(A->comm, 66, "Cannot seek to the column indices");
  ierr = read_ints(viewer->f, lj, (size_t)mine);CHKERRQ(ierr);
  /* the kernels gather x[col] unchecked: a file whose columns are out of range never reaches them */
  for (PetscInt r = 0; r < m; r++) for (PetscInt k = li[r]; k < li[r + 1]; k++)
    if (lj[k] < 0 || lj[k] >= N) SETERRQ(A->comm, 66, "Inconsistant matrix data in file: column %d of row %d is outside [0,%d)", lj[k], rs + r, N);
  if (fseek(viewer->f, vals0 + 8L * before, SEEK_SET)) SETERRQ(A->comm, 66, "Cannot seek to the values");
  ierr = read_scalars(viewer->f, la, (size_t)mine);CHKERRQ(ierr);
  /* MatLoad_SeqAIJ accepts rows whose columns are not in increasing order (aij.c:4093-4157 stores what the file holds); the CSR
   * setters and the kernels here want them sorted, so such a row is sorted with its values (stable insertion: rows are short); a
   * column that appears twice in a row has no meaning in an AIJ file */
  for (PetscInt r = 0; r < m; r++) {
    for (PetscInt k = li[r] + 1; k < li[r + 1]; k++) {
      if (lj[k] > lj[k - 1]) continue;
      const PetscInt c = lj[k]; const PetscScalar v = la[k];
      PetscInt q = k;
      while (q > li[r] && lj[q - 1] > c) { lj[q] = lj[q - 1]; la[q] = la[q - 1]; q--; }
      lj[q] = c; la[q] = v;
    }
    for (PetscInt k = li[r] + 1; k < li[r + 1]; k++)
      if (lj[k] == lj[k - 1]) SETERRQ(A->comm, 66, "Inconsistant matrix data in file: column %d appears twice in row %d", lj[k], rs + r);
  }
  if (fseek(viewer->f, vals0 + 8L * nz, SEEK_SET)) SETERRQ(A->comm, 66, "Cannot seek past the matrix");   /* leave the file positioned after the matrix */
  {   /* MatLoad_SeqAIJ / MatLoad_MPIAIJ hand the rows to the type (aij.c:4140, mpiaij.c:3560): whichever CSR setter it composed */
    PetscVoidFunction fs, fm;
    ierr = PetscObjectQueryFunction((PetscObject)A, "MatSeqAIJSetPreallocationCSR_C", &fs);CHKERRQ(ierr);
    ierr = PetscObjectQueryFunction((PetscObject)A, "MatMPIAIJSetPreallocationCSR_C", &fm);CHKERRQ(ierr);
    if (fs) { ierr = MatSeqAIJSetPreallocationCSR(A, li, lj, la);CHKERRQ(ierr); }
    else if (fm) { ierr = MatMPIAIJSetPreallocationCSR(A, li, lj, la);CHKERRQ(ierr); }
    else SETERRQ(A->comm, PETSC_ERR_SUP, "MatLoad for type %s", A->type_name);
  }
  return 0;
}

PetscErrorCode MatLoad(Mat A, PetscViewer viewer) {
  PetscErrorCode ierr;
  PetscInt header[4], *rowlens = NULL, *li = NULL, *lj = NULL; PetscScalar *la = NULL;
  if (!A || !viewer) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null argument");
  if (viewer->mode != FILE_MODE_READ) SETERRQ(A->comm, PETSC_ERR_ARG_WRONG, "viewer not opened for reading");
  long base = ftell(viewer->f);
  ierr = read_ints(viewer->f, header, 4);CHKERRQ(ierr);
  if (header[0] != MAT_FILE_CLASSID) SETERRQ(A->comm, 79 /* PETSC_ERR_FILE_UNEXPECTED */, "not matrix object");
  if (header[3] < 0) SETERRQ(A->comm, 79, "Matrix stored in special format on disk, cannot load as SeqAIJ");
  if (header[1] < 0 || header[2] < 0) SETERRQ(A->comm, 66 /* PETSC_ERR_FILE_READ */, "Inconsistant matrix data in file: sizes %d x %d", header[1], header[2]);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(header[1], 1), &rowlens);CHKERRQ(ierr);
  ierr = matload_body(A, viewer, base, header, rowlens, &li, &lj, &la);
  free(rowlens); free(li); free(lj); free(la);
  return ierr;
}

PetscErrorCode MatView(Mat A, PetscViewer viewer) {
  PetscErrorCode ierr;
  if (!A || !viewer) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null argument");
  PetscInt m; const PetscInt *ai, *aj; const PetscScalar *aa; PetscVoidFunction f;
  ierr = PetscObjectQueryFunction((PetscObject)A, "MatSeqAIJGetArrays_C", &f);CHKERRQ(ierr);   /* MatView_SeqAIJ_Binary reads the type's CSR arrays */
  if (!f) SETERRQ(A->comm, PETSC_ERR_SUP, "binary MatView is ported for the sequential AIJ type only");
  ierr = ((PetscErrorCode (*)(Mat, PetscInt *, const PetscInt **, const PetscInt **, const PetscScalar **))f)(A, &m, &ai, &aj, &aa);CHKERRQ(ierr);
  if (m > 0 && A->rmap->n != m) {
    /* a blocked type (i, j index bs x bs blocks, stored column-major): MatView_SeqBAIJ_Binary (baij.c:1068-1130) writes the POINT rows
     * -- header {M, N, nnzb bs^2}, every point row's length, its columns bs j + l, its values a[bs^2 k + l bs + r] */
    const PetscInt bs = A->rmap->n / m, bs2 = bs * bs;
    if (bs * m != A->rmap->n || bs < 2) SETERRQ(A->comm, PETSC_ERR_PLIB, "block rows %d do not divide %d rows", m, A->rmap->n);
    const size_t pnz = (size_t)ai[m] * (size_t)bs2;
    PetscInt hdr[4] = {MAT_FILE_CLASSID, A->rmap->n, A->cmap->N, (PetscInt)pnz}, *prl, *pj; PetscScalar *pa;
    if (pnz > 2147483647u) SETERRQ(A->comm, PETSC_ERR_SUP, "matrix too large for the binary format's 32-bit nonzero count");
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)A->rmap->n, &prl);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * PetscMax(pnz, 1), &pj);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscScalar) * PetscMax(pnz, 1), &pa);CHKERRQ(ierr);
    size_t w = 0;
    for (PetscInt I = 0; I < m; I++)
      for (PetscInt r = 0; r < bs; r++) {
        prl[I * bs + r] = bs * (ai[I + 1] - ai[I]);
        for (PetscInt k = ai[I]; k < ai[I + 1]; k++)
          for (PetscInt l = 0; l < bs; l++) { pj[w] = bs * aj[k] + l; pa[w] = aa[(size_t)bs2 * (size_t)k + (size_t)l * bs + r]; w++; }
      }
    ierr = write_ints(viewer->f, hdr, 4);
    if (!ierr) ierr = write_ints(viewer->f, prl, (size_t)A->rmap->n);
    if (!ierr) ierr = write_ints(viewer->f, pj, pnz);
    if (!ierr) ierr = write_scalars(viewer->f, pa, pnz);
    free(prl); free(pj); free(pa);
    CHKERRQ(ierr);
    return 0;
  }
  PetscInt header[4] = {MAT_FILE_CLASSID, m, A->cmap->N, ai[m]}, *rl;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(m, 1), &rl);CHKERRQ(ierr);
  for (PetscInt r = 0; r < m; r++) rl[r] = ai[r + 1] - ai[r];
  ierr = write_ints(viewer->f, header, 4);CHKERRQ(ierr);
  ierr = write_ints(viewer->f, rl, (size_t)m);CHKERRQ(ierr);
  ierr = write_ints(viewer->f, aj, (size_t)ai[m]);CHKERRQ(ierr);
  ierr = write_scalars(viewer->f, aa, (size_t)ai[m]);CHKERRQ(ierr);
  free(rl);
  return 0;
}

PetscErrorCode VecLoad(Vec v, PetscViewer viewer) {   /* src/vec/vec/utils/vecio.c */
  PetscErrorCode ierr;
  PetscInt header[2];
  if (!v || !viewer) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null argument");
  long base = ftell(viewer->f);
  ierr = read_ints(viewer->f, header, 2);CHKERRQ(ierr);
  if (header[0] != VEC_FILE_CLASSID) SETERRQ(v->comm, 79, "Not vector next in file");
  if (!v->map) { ierr = VecSetSizes(v, PETSC_DECIDE, header[1]);CHKERRQ(ierr); }
  if (!v->type_name[0]) { ierr = VecSetType(v, VECSTANDARD);CHKERRQ(ierr); }
  if (v->map->N != header[1]) SETERRQ(v->comm, 79, "Vector in file different length (%d) then input vector (%d)", header[1], v->map->N);
  PetscScalar *a;
  ierr = VecGetArray(v, &a);CHKERRQ(ierr);
  fseek(viewer->f, base + 8 + 8L * v->map->rstart, SEEK_SET);
  ierr = read_scalars(viewer->f, a, (size_t)v->map->n);CHKERRQ(ierr);
  ierr = VecRestoreArray(v, &a);CHKERRQ(ierr);
  fseek(viewer->f, base + 8 + 8L * header[1], SEEK_SET);
  return 0;
}

PetscErrorCode VecView(Vec v, PetscViewer viewer) {
  PetscErrorCode ierr;
  if (!v || !viewer) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null argument");
  if (v->comm->size > 1) SETERRQ(v->comm, PETSC_ERR_SUP, "binary VecView is ported for sequential vectors only");
  PetscInt header[2] = {VEC_FILE_CLASSID, v->map->N};
  const PetscScalar *a;
  ierr = write_ints(viewer->f, header, 2);CHKERRQ(ierr);
  ierr = VecGetArrayRead(v, &a);CHKERRQ(ierr);
  ierr = write_scalars(viewer->f, a, (size_t)v->map->n);CHKERRQ(ierr);
  ierr = VecRestoreArrayRead(v, &a);CHKERRQ(ierr);
  return 0;
}

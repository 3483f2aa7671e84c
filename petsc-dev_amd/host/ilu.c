/* PCILU with zero fill on a sequential AIJ block: the reference's default preconditioner on one rank
 * and its default sub-preconditioner under PCBJACOBI (src/ksp/pc/interface/precon.c:14-53).
 *   set-up : MatILUFactorSymbolic_SeqAIJ_ilu0 + MatLUFactorNumeric_SeqAIJ (src/mat/impls/aij/seq/aijfact.c:1628, :461),
 *            natural ordering, on the HOST copy of the matrix -- as the reference's own GPU back end does
 *            (src/mat/impls/aij/seq/seqcusparse/aijcusparse.cu:192-445 factors on the CPU and solves on the GPU);
 *            then a dependency-level analysis of L and U and one upload.
 *   apply  : MatSolve_SeqAIJ_NaturalOrdering (aijfact.c:3126) on the device, one lane per row in column order (same bits):
 *            by default two launches, one per triangular solve, with point-to-point hand-off of the solution values
 *            between wavefronts (mi355x_trisolve_*, csrc/trisolve.hip); -pc_factor_hipmi355x_trisolve level selects the
 *            level-scheduled kernels, one launch per dependency level (replayed from a hipGraph), which also serve
 *            systems with few levels and as the fall-back. */
#include "hipmi355ximpl.h"

typedef struct {
  PetscInt n, nz;
  PetscInt *bi, *bj, *bdiag; PetscScalar *ba;            /* host factors */
  PetscInt *d_bi, *d_bj, *d_bdiag; PetscScalar *d_ba;    /* device copies */
  PetscInt nlevL, nlevU, *levptrL, *levptrU;             /* level pointers (host) */
  PetscInt *d_rowsL, *d_rowsU;                           /* rows ordered by level (device) */
  PetscScalar *d_work;                                   /* fixed in-place buffer the captured graph works on */
  void *graph;                                           /* hipGraphExec of the nlevL + nlevU level launches */
  int graph_tried;
  mi355x_trisolve_plan_t tri_lo, tri_up;                 /* sync-free solves (NULL: level launches) */
  int by_level;                                          /* rows summed in dependency-level order (inode matrices) instead of column order */
  PetscInt nshift;                                       /* restarts of the factorisation MatPivotCheck_nz asked for (largest count over the blocks) */
  int factored_state;
  PetscInt nblk, *blk;                                   /* "PCFactorSetIndependentBlocks_C": the matrix stands for that many separate matrices (row ranges), each with its own shift loop */
} PC_ILU;

static PetscErrorCode ilu_free(PC_ILU *f) {
  HipFree(f->bi); HipFree(f->bj); HipFree(f->bdiag); HipFree(f->ba); HipFree(f->levptrL); HipFree(f->levptrU);
  if (f->d_bi) mi355x_free(f->d_bi);
  if (f->d_bj) mi355x_free(f->d_bj);
  if (f->d_bdiag) mi355x_free(f->d_bdiag);
  if (f->d_ba) mi355x_free(f->d_ba);
  if (f->d_rowsL) mi355x_free(f->d_rowsL);
  if (f->d_rowsU) mi355x_free(f->d_rowsU);
  if (f->d_work) mi355x_free(f->d_work);
  if (f->graph) mi355x_graph_destroy(f->graph);
  if (f->tri_lo) mi355x_trisolve_plan_destroy(f->tri_lo);
  if (f->tri_up) mi355x_trisolve_plan_destroy(f->tri_up);
  { PetscInt nblk = f->nblk, *blk = f->blk;
    memset(f, 0, sizeof(*f));
    f->nblk = nblk; f->blk = blk; }
  f->factored_state = -1;
  return 0;
}

/* block Jacobi solving all its ILU(0) blocks as one block-diagonal system: every block is factored as the reference factors a
 * matrix of its own (its own restarts), so the result is the blocks' factors side by side also when one block needs a shift */
static PetscErrorCode PCFactorSetIndependentBlocks_ILU(PC pc, PetscInt nblk, const PetscInt *starts) {
  PC_ILU *f = (PC_ILU *)pc->data;
  PetscErrorCode ierr;
  HipFree(f->blk); f->blk = NULL; f->nblk = 0;
  if (nblk > 0) {
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nblk + 1), &f->blk);CHKERRQ(ierr);
    memcpy(f->blk, starts, sizeof(PetscInt) * (size_t)(nblk + 1));
    f->nblk = nblk;
  }
  f->factored_state = -1;
  return 0;
}

/* rows sorted by dependency level (stable: ascending row inside a level) */
static PetscErrorCode level_order(PetscInt n, const PetscInt *lev, PetscInt nlev, PetscInt **ptr_out, PetscInt **rows_out) {
  PetscErrorCode ierr;
  PetscInt *ptr, *rows, *next;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nlev + 1), &ptr);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &rows);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nlev + 1), &next);CHKERRQ(ierr);
  memset(ptr, 0, sizeof(PetscInt) * (size_t)(nlev + 1));
  for (PetscInt i = 0; i < n; i++) ptr[lev[i] + 1]++;
  for (PetscInt l = 0; l < nlev; l++) ptr[l + 1] += ptr[l];
  memcpy(next, ptr, sizeof(PetscInt) * (size_t)(nlev + 1));
  for (PetscInt i = 0; i < n; i++) rows[next[lev[i]]++] = i;
  HipFree(next);
  *ptr_out = ptr; *rows_out = rows;
  return 0;
}

static PetscErrorCode PCSetUp_ILU(PC pc) {
  PetscErrorCode ierr;
  PC_ILU *f = (PC_ILU *)pc->data;
  Mat A = pc->pmat;
  PetscInt n; const PetscInt *ai, *aj; const PetscScalar *aa;
  PetscDeviceCtx *dc;
  if (strcmp(HipObjTypeName(A), MATSEQAIJHIPMI355X)) SETERRQ(HipObjComm(pc), PETSC_ERR_SUP, "PCILU needs a sequential AIJ matrix (use -pc_type bjacobi -sub_pc_type ilu in parallel); got %s", HipObjTypeName(A));
  if (f->factored_state == HipObjState(A) && f->d_ba) return 0;
  ierr = MatSeqAIJGetArrays(A, &n, &ai, &aj, &aa);CHKERRQ(ierr);
  if (A->rmap->n != A->cmap->n) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONG, "Must be square matrix, rows %d columns %d", A->rmap->n, A->cmap->n);
  ierr = ilu_free(f);CHKERRQ(ierr);
  f->n = n; f->nz = ai[n];
  PetscInt *adiag;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &adiag);CHKERRQ(ierr);
  for (PetscInt i = 0; i < n; i++) {
    adiag[i] = -1;
    for (PetscInt q = ai[i]; q < ai[i + 1]; q++) if (aj[q] == i) { adiag[i] = q; break; }
    if (adiag[i] < 0) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONGSTATE, "Matrix is missing diagonal entry %d", i);
  }
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(n + 1), &f->bi);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(f->nz + 1), &f->bj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(n + 1), &f->bdiag);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)(f->nz + 1), &f->ba);CHKERRQ(ierr);
  memset(f->ba, 0, sizeof(PetscScalar) * (size_t)(f->nz + 1));
  /* symbolic: the pattern of A, L part forward, U part from the last row backwards (aijfact.c:1660-1685) */
  PetscInt k = 0, *bi = f->bi, *bj = f->bj, *bdiag = f->bdiag; PetscScalar *ba = f->ba;
  bi[0] = 0;
  for (PetscInt i = 0; i < n; i++) {
    PetscInt nzl = adiag[i] - ai[i];
    bi[i + 1] = bi[i] + nzl;
    for (PetscInt j = 0; j < nzl; j++) bj[k++] = aj[ai[i] + j];
  }
  bdiag[n] = bi[n] - 1;
  for (PetscInt i = n - 1; i >= 0; i--) {
    PetscInt nzu = ai[i + 1] - adiag[i] - 1;
    for (PetscInt j = 0; j < nzu; j++) bj[k++] = aj[adiag[i] + 1 + j];
    bj[k++] = i;
    bdiag[i] = bdiag[i + 1] + nzu + 1;
  }
  /* numeric (aijfact.c:505-570): row by row with a dense work row; pivots are stored inverted */
  PetscScalar *rtmp;
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)(n + 1), &rtmp);CHKERRQ(ierr);
  const PetscReal zeropivot = 100.0 * 2.220446049250313e-16, shiftamount = 100.0 * 2.220446049250313e-16;   /* ilu.c:388-389 */
  f->nshift = 0;
  const PetscInt whole[2] = {0, n};
  const PetscInt nblk = (f->nblk > 0 && f->blk[f->nblk] == n) ? f->nblk : 1, *blk = (f->nblk > 0 && f->blk[f->nblk] == n) ? f->blk : whole;
  for (PetscInt bb = 0; bb < nblk; bb++) {
    const PetscInt r0 = blk[bb], r1 = blk[bb + 1];
    for (PetscInt i = r0; i < r1; i++) {
      if (ai[i] < ai[i + 1] && (aj[ai[i]] < r0 || aj[ai[i + 1] - 1] >= r1)) { HipFree(rtmp); HipFree(adiag); SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONG, "row %d couples to a column outside its independent block", i); }
    }
    PetscReal shift_amount = 0.0;
    PetscInt nshift = 0;
    PetscBool again;
    do {   /* MAT_SHIFT_NONZERO, PCILU's default on a SeqAIJ matrix (ilu.c:387): a pivot that fails MatPivotCheck_nz (matimpl.h:512-528)
            * restarts the factorisation with the diagonal shifted by shiftamount, then by twice that, ... (aijfact.c:507-592) */
      again = PETSC_FALSE;
      for (PetscInt i = r0; i < r1; i++) {
        PetscInt nzl = bi[i + 1] - bi[i], nzu = bdiag[i] - bdiag[i + 1];
        PetscReal rs = 0.0;
        for (PetscInt j = 0; j < nzl; j++) rtmp[bj[bi[i] + j]] = 0.0;
        for (PetscInt j = 0; j < nzu; j++) rtmp[bj[bdiag[i + 1] + 1 + j]] = 0.0;
        for (PetscInt q = ai[i]; q < ai[i + 1]; q++) rtmp[aj[q]] = aa[q];
        rtmp[i] += shift_amount;
        for (PetscInt kk = 0; kk < nzl; kk++) {
          const PetscInt row = bj[bi[i] + kk];
          PetscScalar *pc_ = rtmp + row;
          if (*pc_ != 0.0) {
            const PetscScalar multiplier = *pc_ * ba[bdiag[row]];
            *pc_ = multiplier;
            const PetscInt *pj = bj + bdiag[row + 1] + 1;
            const PetscScalar *pv = ba + bdiag[row + 1] + 1;
            const PetscInt nz = bdiag[row] - bdiag[row + 1] - 1;
            for (PetscInt j = 0; j < nz; j++) rtmp[pj[j]] -= multiplier * pv[j];
          }
        }
        for (PetscInt j = 0; j < nzl; j++) { ba[bi[i] + j] = rtmp[bj[bi[i] + j]]; rs += PetscAbsScalar(ba[bi[i] + j]); }
        for (PetscInt j = 0; j < nzu - 1; j++) { ba[bdiag[i + 1] + 1 + j] = rtmp[bj[bdiag[i + 1] + 1 + j]]; rs += PetscAbsScalar(ba[bdiag[i + 1] + 1 + j]); }
        if (PetscAbsScalar(rtmp[i]) <= zeropivot * rs) {
          shift_amount = nshift ? shift_amount * 2.0 : shiftamount;
          nshift++;
          if (nshift > 80) { HipFree(rtmp); HipFree(adiag); SETERRQ(HipObjComm(pc), 71 /* PETSC_ERR_MAT_LU_ZRPVT */, "Zero pivot row %d value %g: still there after %d diagonal shifts", i, PetscAbsScalar(rtmp[i]), nshift); }
          again = PETSC_TRUE;
          break;
        }
        ba[bdiag[i]] = 1.0 / rtmp[i];
      }
    } while (again);
    f->nshift = PetscMax(f->nshift, nshift);
  }
  HipFree(rtmp); HipFree(adiag);
  /* dependency levels: a row may start once the rows it references are done */
  PetscInt *lev, *levU, *rowsL, *rowsU;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &lev);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &levU);CHKERRQ(ierr);
  f->nlevL = 0;
  for (PetscInt i = 0; i < n; i++) {
    PetscInt l = 0;
    for (PetscInt q = bi[i]; q < bi[i + 1]; q++) l = PetscMax(l, lev[bj[q]] + 1);
    lev[i] = l; f->nlevL = PetscMax(f->nlevL, l + 1);
  }
  ierr = level_order(n, lev, f->nlevL, &f->levptrL, &rowsL);CHKERRQ(ierr);
  f->nlevU = 0;
  for (PetscInt i = n - 1; i >= 0; i--) {
    PetscInt l = 0, s0 = bdiag[i + 1] + 1, nz = bdiag[i] - bdiag[i + 1] - 1;
    for (PetscInt q = 0; q < nz; q++) l = PetscMax(l, levU[bj[s0 + q]] + 1);
    levU[i] = l; f->nlevU = PetscMax(f->nlevU, l + 1);
  }
  ierr = level_order(n, levU, f->nlevU, &f->levptrU, &rowsU);CHKERRQ(ierr);
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  {   /* sync-free solves: worth it as soon as the level launches would be a launch-bound chain */
    char mode[32] = "syncfree"; PetscBool set;
    ierr = PetscOptionsGetString(HipObjPrefix(pc), "-pc_factor_hipmi355x_trisolve", mode, sizeof(mode), &set);CHKERRQ(ierr);
    if (strcmp(mode, "syncfree") && strcmp(mode, "level")) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONG, "-pc_factor_hipmi355x_trisolve <syncfree|level>, got %s", mode);
    if (!strcmp(mode, "syncfree") && n > 0 && (f->nlevL + f->nlevU > 16 || set)) {
      PetscInt *rpU, *rlU, *rlL; PetscScalar *dinv;
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &rpU);CHKERRQ(ierr);
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &rlU);CHKERRQ(ierr);
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &rlL);CHKERRQ(ierr);
      ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)n, &dinv);CHKERRQ(ierr);
      for (PetscInt i = 0; i < n; i++) {
        rlL[i] = bi[i + 1] - bi[i];
        rpU[i] = bdiag[i + 1] + 1; rlU[i] = bdiag[i] - bdiag[i + 1] - 1; dinv[i] = ba[bdiag[i]];
      }
      /* -pc_factor_hipmi355x_trisolve_order <column|level>.  column: every row is summed in column order, the bits of
       * MatSolve_SeqAIJ_NaturalOrdering.  level: in the order of its dependencies' levels (a row then waits on its last
       * entries only) -- the default where the reference does not run the natural-ordering routine either: a matrix with
       * inodes, whose factor it solves with MatSolve_SeqAIJ_Inode (inode.c; MatLUFactorNumeric_SeqAIJ_Inode installs it),
       * in yet another order.  There the two agree to rounding. */
      char ord[32] = ""; PetscInt nodes = 0; int by_level;
      ierr = PetscOptionsGetString(HipObjPrefix(pc), "-pc_factor_hipmi355x_trisolve_order", ord, sizeof(ord), &set);CHKERRQ(ierr);
      if (set && strcmp(ord, "column") && strcmp(ord, "level")) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONG, "-pc_factor_hipmi355x_trisolve_order <column|level>, got %s", ord);
      ierr = MatHIPMI355XGetInodeInfo(A, &nodes, NULL, NULL);CHKERRQ(ierr);
      by_level = set ? !strcmp(ord, "level") : (nodes > 0);
      f->by_level = by_level;
      int rc = mi355x_trisolve_plan_create_ordered(dc->h, n, f->nlevL, lev, bi, rlL, bj, ba, NULL, by_level, &f->tri_lo);
      if (!rc) rc = mi355x_trisolve_plan_create_ordered(dc->h, n, f->nlevU, levU, rpU, rlU, bj, ba, dinv, by_level, &f->tri_up);
      HipFree(rpU); HipFree(rlU); HipFree(rlL); HipFree(dinv);
      if (rc) {   /* e.g. a factor too large for 32-bit sliced-ELL offsets: the level kernels serve */
        if (f->tri_lo) mi355x_trisolve_plan_destroy(f->tri_lo);
        if (f->tri_up) mi355x_trisolve_plan_destroy(f->tri_up);
        f->tri_lo = f->tri_up = NULL;
      }
    }
  }
  HipFree(lev); HipFree(levU);
  CHKHIP(mi355x_malloc((void **)&f->d_bi, sizeof(PetscInt) * (size_t)(n + 1)));
  CHKHIP(mi355x_malloc((void **)&f->d_bj, sizeof(PetscInt) * (size_t)(f->nz + 1)));
  CHKHIP(mi355x_malloc((void **)&f->d_bdiag, sizeof(PetscInt) * (size_t)(n + 1)));
  CHKHIP(mi355x_malloc((void **)&f->d_ba, sizeof(PetscScalar) * (size_t)(f->nz + 1)));
  CHKHIP(mi355x_malloc((void **)&f->d_rowsL, sizeof(PetscInt) * (size_t)PetscMax(n, 1)));
  CHKHIP(mi355x_malloc((void **)&f->d_rowsU, sizeof(PetscInt) * (size_t)PetscMax(n, 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_bi, bi, sizeof(PetscInt) * (size_t)(n + 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_bj, bj, sizeof(PetscInt) * (size_t)(f->nz + 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_bdiag, bdiag, sizeof(PetscInt) * (size_t)(n + 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_ba, ba, sizeof(PetscScalar) * (size_t)(f->nz + 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_rowsL, rowsL, sizeof(PetscInt) * (size_t)n));
  CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_rowsU, rowsU, sizeof(PetscInt) * (size_t)n));
  CHKHIP(mi355x_handle_synchronize(dc->h));
  HipFree(rowsL); HipFree(rowsU);
  f->factored_state = HipObjState(A);
  return 0;
}

static PetscErrorCode PCApply_ILU(PC pc, Vec x, Vec y) {   /* PCApply_ILU -> MatSolve(fact, x, y) */
  PetscErrorCode ierr;
  PC_ILU *f = (PC_ILU *)pc->data;
  const PetscScalar *db; PetscScalar *dx; PetscDeviceCtx *dc;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = VecHIPGetRead(x, &db);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(y, &dx);CHKERRQ(ierr);
  /* The level launches are a launch-bound inner loop (766 + 766 kernels for P7(256)): with more than a handful of
   * levels they are captured once into a hipGraph that works in place on a fixed buffer and replayed per
   * application (copy in, one graph launch, copy out).  Same kernels, same order, same bits. */
  if (f->tri_lo) {
    int rc = mi355x_trisolve_apply(dc->h, f->tri_lo, f->tri_up, db, dx);
    if (rc == 719) {   /* hipErrorLaunchFailure: an earlier application gave up on a dependency (its result was unusable) */
      mi355x_trisolve_plan_destroy(f->tri_lo); mi355x_trisolve_plan_destroy(f->tri_up);
      f->tri_lo = f->tri_up = NULL;
      SETERRQ(HipObjComm(pc), PETSC_ERR_LIB, "sync-free triangular solve timed out in an earlier application; the level-scheduled solves are used from now on");
    }
    CHKHIP(rc);
    ierr = VecHIPRestoreWrite(y);CHKERRQ(ierr);
    HipStateIncrease(y);
    ierr = PetscLogFlops(2.0 * f->nz - f->n);CHKERRQ(ierr);
    return 0;
  }
  if (!f->graph_tried && f->nlevL + f->nlevU > 16) {
    f->graph_tried = 1;
    if (!mi355x_malloc((void **)&f->d_work, sizeof(PetscScalar) * (size_t)PetscMax(f->n, 1)) && !mi355x_graph_capture_begin(dc->h)) {
      int bad = 0;
      for (PetscInt l = 0; l < f->nlevL && !bad; l++)
        bad = mi355x_ilu0_lower_level(dc->h, f->levptrL[l + 1] - f->levptrL[l], f->d_rowsL + f->levptrL[l], f->d_bi, f->d_bj, f->d_ba, f->d_work, f->d_work);
      for (PetscInt l = 0; l < f->nlevU && !bad; l++)
        bad = mi355x_ilu0_upper_level(dc->h, f->levptrU[l + 1] - f->levptrU[l], f->d_rowsU + f->levptrU[l], f->d_bj, f->d_ba, f->d_bdiag, f->d_work);
      void *g = NULL;
      if (mi355x_graph_capture_end(dc->h, &g) || bad) g = NULL;   /* capture failed: stay with plain launches */
      f->graph = g;
    }
  }
  if (f->graph) {
    CHKHIP(mi355x_vec_copy(dc->h, (size_t)f->n, db, f->d_work));
    CHKHIP(mi355x_graph_launch(dc->h, f->graph));
    CHKHIP(mi355x_vec_copy(dc->h, (size_t)f->n, f->d_work, dx));
  } else {
    for (PetscInt l = 0; l < f->nlevL; l++)
      CHKHIP(mi355x_ilu0_lower_level(dc->h, f->levptrL[l + 1] - f->levptrL[l], f->d_rowsL + f->levptrL[l], f->d_bi, f->d_bj, f->d_ba, db, dx));
    for (PetscInt l = 0; l < f->nlevU; l++)
      CHKHIP(mi355x_ilu0_upper_level(dc->h, f->levptrU[l + 1] - f->levptrU[l], f->d_rowsU + f->levptrU[l], f->d_bj, f->d_ba, f->d_bdiag, dx));
  }
  ierr = VecHIPRestoreWrite(y);CHKERRQ(ierr);
  HipStateIncrease(y);
  ierr = PetscLogFlops(2.0 * f->nz - f->n);CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode PCDestroy_ILU(PC pc) {
  PC_ILU *f = (PC_ILU *)pc->data;
  if (f) { ilu_free(f); HipFree(f->blk); HipFree(f); pc->data = NULL; }
  (void)PetscObjectComposeFunction((PetscObject)pc, "PCFactorSetIndependentBlocks_C", "", (PetscVoidFunction)NULL);
  return 0;
}

PetscErrorCode PCCreate_ILU_HIPMI355X(PC pc) {
  PC_ILU *f;
  PetscErrorCode ierr = PetscMalloc(sizeof(*f), &f);CHKERRQ(ierr);
  memset(f, 0, sizeof(*f));
  f->factored_state = -1;
  pc->data = f;
  pc->ops->setup = PCSetUp_ILU; pc->ops->apply = PCApply_ILU; pc->ops->destroy = PCDestroy_ILU;
  ierr = PetscObjectComposeFunction((PetscObject)pc, "PCFactorSetIndependentBlocks_C", "PCFactorSetIndependentBlocks_ILU", (PetscVoidFunction)PCFactorSetIndependentBlocks_ILU);CHKERRQ(ierr);
  return 0;
}

/* 1 when PCApply runs the sync-free solves (two launches), 0 for the level-scheduled kernels; *aborted: a dependency wait gave up */
PetscErrorCode PCILUGetSolver_HIPMI355X(PC pc, PetscInt *syncfree, PetscInt *aborted) {
  if (strcmp(HipObjTypeName(pc), "ilu")) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONG, "not a PCILU");
  PC_ILU *f = (PC_ILU *)pc->data;
  int a = 0, b = 0;
  if (syncfree) *syncfree = f->tri_lo ? 1 : 0;
  if (f->tri_lo) {
    PetscDeviceCtx *dc;
    PetscErrorCode ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
    CHKHIP(mi355x_handle_synchronize(dc->h));      /* the flag of everything queued so far */
    mi355x_trisolve_aborted(f->tri_lo, &a); mi355x_trisolve_aborted(f->tri_up, &b);
  }
  if (aborted) *aborted = a || b;
  return 0;
}

/* restarts of the factorisation with a larger diagonal shift (MAT_SHIFT_NONZERO); 0 for every matrix whose pivots pass */
PetscErrorCode PCILUGetShiftCount_HIPMI355X(PC pc, PetscInt *nshift) {
  if (strcmp(HipObjTypeName(pc), "ilu") && strcmp(HipObjTypeName(pc), "iluhipmi355x")) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONG, "not a PCILU");
  *nshift = ((PC_ILU *)pc->data)->nshift;
  return 0;
}

/* levels of the two triangular solves (for tests / DESIGN.md) */
PetscErrorCode PCILUGetLevels_HIPMI355X(PC pc, PetscInt *nlevL, PetscInt *nlevU) {
  if (strcmp(HipObjTypeName(pc), "ilu")) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONG, "not a PCILU");
  PC_ILU *f = (PC_ILU *)pc->data;
  if (nlevL) *nlevL = f->nlevL;
  if (nlevU) *nlevU = f->nlevU;
  return 0;
}

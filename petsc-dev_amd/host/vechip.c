/* VECSEQHIPMI355X / VECMPIHIPMI355X: vectors resident in HBM.  The role of
 * src/vec/vec/impls/seq/seqcusp/veccusp.cu and src/vec/vec/impls/mpi/mpicusp/mpicusp.cu in the
 * reference (ops tables veccusp.cu:1915-1941, mpicusp.cu:193-217), written against the C ABI of
 * mi355x_kernels.h.  Coherence: a host mirror exists only after the first host access; flags as in
 * cuspvecimpl.h:95-150.  Reductions: device two-level tree -> (RCCL all-reduce when the
 * communicator has more than one rank, replacing MPI_Allreduce of pbvec.c:16,30 / pvec2.c:20,62-80)
 * -> pinned host scalar -> one stream synchronise. */
#include "hipmi355ximpl.h"

#define VH(v) ((Vec_HIPMI355X *)(v)->data)
PetscErrorCode VecCGUpdateCheck_HIPMI355X(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscBool *ok);

static PetscErrorCode dev_alloc(Vec v) {
  Vec_HIPMI355X *s = VH(v);
  if (!s->dev) {
    PetscDeviceCtx *dc;
    PetscErrorCode ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
    CHKHIP(mi355x_malloc((void **)&s->dev, sizeof(PetscScalar) * (size_t)PetscMax(v->map->n, 2)));
    if (s->valid == VALID_NONE) {   /* new vectors are zero, as VecCreate_Seq's PetscMemzero */
      CHKHIP(mi355x_memset(dc->h, s->dev, 0, sizeof(PetscScalar) * (size_t)v->map->n));
      s->valid = VALID_DEVICE;
    }
  }
  return 0;
}
static PetscErrorCode host_alloc(Vec v) {
  Vec_HIPMI355X *s = VH(v);
  if (!s->host) {
    PetscErrorCode ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(v->map->n, 1), &s->host);CHKERRQ(ierr);
    s->host_owned = 1;
    if (s->valid == VALID_NONE) { memset(s->host, 0, sizeof(PetscScalar) * (size_t)v->map->n); s->valid = VALID_HOST; }
  }
  return 0;
}
static PetscErrorCode to_device(Vec v) {   /* VecCUSPCopyToGPU, veccusp.cu:109 */
  Vec_HIPMI355X *s = VH(v);
  PetscErrorCode ierr = dev_alloc(v);CHKERRQ(ierr);
  if (s->valid == VALID_HOST) {
    PetscDeviceCtx *dc;
    ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
    CHKHIP(mi355x_memcpy_h2d(dc->h, s->dev, s->host, sizeof(PetscScalar) * (size_t)v->map->n));
    CHKHIP(mi355x_handle_synchronize(dc->h));   /* the host buffer is pageable */
    s->valid = VALID_BOTH;
  }
  return 0;
}
static PetscErrorCode to_host(Vec v) {   /* VecCUSPCopyFromGPU, veccusp.cu:173 */
  Vec_HIPMI355X *s = VH(v);
  PetscErrorCode ierr = host_alloc(v);CHKERRQ(ierr);
  if (s->valid == VALID_DEVICE) {
    PetscDeviceCtx *dc;
    ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
    CHKHIP(mi355x_memcpy_d2h(dc->h, s->host, s->dev, sizeof(PetscScalar) * (size_t)v->map->n));
    CHKHIP(mi355x_handle_synchronize(dc->h));
    ierr = HipTriWatchCheck();CHKERRQ(ierr);     /* did a sync-free triangular solve queued before this copy give up?  (host/ilu.c) */
    s->valid = VALID_BOTH;
  }
  return 0;
}

/* ---- deferred CG sweep (the queue and what it is for: further down, "deferred element-wise operations") ---- */
enum { DQ_AXPY = 1, DQ_PMULT, DQ_COPY, DQ_AXPBYPCZ, DQ_WAXPY, DQ_MAXPY };
typedef struct { int kind; Vec o, a, b; PetscScalar s1, s2, s3; } DqOp;   /* o: the vector written; a, b: read; the call's scalars */
static struct {
  int n, busy;                      /* noted operations (0..3), a prefix of one of the patterns below; busy: the note is being run */
  DqOp op[3];
  Vec ca, cb; long sa, sb; PetscScalar cval; int have;   /* a dot product (ca, cb) left by a fused sweep, valid while both keep these states */
  PetscInt mnv; PetscScalar malpha[32]; Vec mx[32];   /* DQ_MAXPY (op[0].o += sum malpha[j] mx[j]): GMRES's update, run inside the VecNorm behind it */
  Vec la[4], lb[4]; int lpos;       /* operands of the last few VecDot / VecTDot calls (BiCGStab: (r, rp) of the iteration's start names the partner the fused update needs) */
} dq;
/* a product noted by the Mat type (MatMult of a sequential AIJ matrix, "a noted product" below).  pp: t = A x noted, nothing launched.
 * pl: d .* (A x) has been launched into another vector and t itself is still unwritten (written only if somebody needs it). */
typedef struct { int on; Mat A; long astate; Vec x, t; HipProductNowFn now; HipProductScaledFn scaled; } PendingProduct;
static PendingProduct pp, pl;
static int pp_busy;
static PetscErrorCode product_run(PendingProduct *q);
static PetscErrorCode deferred_flush(void);
#define FLUSH_DEFERRED() do { if ((dq.n && !dq.busy) || (pp.on && !pp_busy)) { PetscErrorCode e__ = deferred_flush();CHKERRQ(e__); } } while (0)
/* the unwritten t of pl: a reader gets it written first; a writer of t makes it dead; a writer of x gets it written first */
#define PP_READ(v)  do { if (pl.on && !pp_busy && (v) == pl.t) { PetscErrorCode e__ = product_run(&pl);CHKERRQ(e__); } } while (0)
#define PP_WRITE(v) do { if (pl.on && !pp_busy) { if ((v) == pl.t) pl.on = 0; else if ((v) == pl.x) { PetscErrorCode e__ = product_run(&pl);CHKERRQ(e__); } } } while (0)
#define PP_RW(v)    do { if (pl.on && !pp_busy && ((v) == pl.t || (v) == pl.x)) { PetscErrorCode e__ = product_run(&pl);CHKERRQ(e__); } } while (0)

PetscErrorCode VecHIPGetRead(Vec v, const PetscScalar **d) {
  FLUSH_DEFERRED();
  PP_READ(v);
  PetscErrorCode ierr = to_device(v);CHKERRQ(ierr);
  *d = VH(v)->dev;
  return 0;
}
PetscErrorCode VecHIPGetWrite(Vec v, PetscScalar **d) {
  FLUSH_DEFERRED();
  PP_WRITE(v);
  PetscErrorCode ierr = dev_alloc(v);CHKERRQ(ierr);
  *d = VH(v)->dev;
  return 0;
}
PetscErrorCode VecHIPGetReadWrite(Vec v, PetscScalar **d) {
  FLUSH_DEFERRED();
  PP_RW(v);
  PetscErrorCode ierr = to_device(v);CHKERRQ(ierr);
  *d = VH(v)->dev;
  return 0;
}
PetscErrorCode VecHIPRestoreWrite(Vec v) { VH(v)->valid = VALID_DEVICE; return 0; }

static int is_hip(Vec v) { return v && v->data && strstr(HipObjTypeName(v), "hipmi355x") != NULL; }
#define CheckHIP(v) do { if (!is_hip(v)) SETERRQ(HipObjComm(v), PETSC_ERR_ARG_NOTSAMETYPE, "vector of type %s mixed with a HIPMI355X vector", HipObjTypeName(v)); } while (0)

PetscErrorCode VecHIPMI355XGetArray(Vec v, PetscScalar **d) { CheckHIP(v); return VecHIPGetReadWrite(v, d); }
PetscErrorCode VecHIPMI355XRestoreArray(Vec v, PetscScalar **d) { if (d) *d = NULL; VecHIPRestoreWrite(v); HipStateIncrease(v); return 0; }
PetscErrorCode VecHIPMI355XGetArrayRead(Vec v, const PetscScalar **d) { CheckHIP(v); return VecHIPGetRead(v, d); }

/* ---- host access ---- */
static PetscErrorCode VecGetArray_HIP(Vec v, PetscScalar **a) {
  FLUSH_DEFERRED();
  PP_RW(v);
  PetscErrorCode ierr = to_host(v);CHKERRQ(ierr);
  *a = VH(v)->host;
  return 0;
}
static PetscErrorCode VecRestoreArray_HIP(Vec v, PetscScalar **a) { if (a) *a = NULL; VH(v)->valid = VALID_HOST; return 0; }
/* VecPlaceArray_SeqCUSP (veccusp.cu): adopt a host array; the device copy is refreshed on next use */
static PetscErrorCode VecPlaceArray_HIP(Vec v, const PetscScalar *a) {
  Vec_HIPMI355X *s = VH(v);
  if (s->placed_save) SETERRQ(HipObjComm(v), PETSC_ERR_ARG_WRONGSTATE, "VecPlaceArray() was already called on this vector, without a call to VecResetArray()");
  FLUSH_DEFERRED();
  PP_RW(v);
  PetscErrorCode ierr = to_host(v);CHKERRQ(ierr);
  s->placed_save = s->host;
  s->host = (PetscScalar *)a;
  s->valid = VALID_HOST;
  return 0;
}
/* VecReplaceArray_Seq (dvec2.c:1159-1168): the vector's own host array is freed and `a` takes its place for good -- the vector
 * owns it from now on (allocated with PetscMalloc, as the reference requires) -- and holds the current values */
static PetscErrorCode VecReplaceArray_HIP(Vec v, const PetscScalar *a) {
  Vec_HIPMI355X *s = VH(v);
  FLUSH_DEFERRED();
  PP_RW(v);
  if (s->placed_save) { if (s->host_owned) HipFree(s->placed_save); s->placed_save = NULL; }   /* the saved array was the one this vector allocated */
  else if (s->host && s->host_owned) HipFree(s->host);
  s->host = (PetscScalar *)a;
  s->host_owned = 1;
  s->valid = VALID_HOST;
  return 0;
}
static PetscErrorCode VecResetArray_HIP(Vec v) {
  Vec_HIPMI355X *s = VH(v);
  if (!s->placed_save) return 0;
  FLUSH_DEFERRED();
  PP_RW(v);
  s->host = s->placed_save;
  s->placed_save = NULL;
  s->valid = VALID_HOST;
  return 0;
}
static PetscErrorCode VecSetValues_HIP(Vec v, PetscInt ni, const PetscInt ix[], const PetscScalar y[], InsertMode mode) {
  PetscScalar *a = NULL;
  PetscErrorCode ierr = VecGetArray_HIP(v, &a);CHKERRQ(ierr);
  for (PetscInt k = 0; k < ni; k++) {
    if (ix[k] < 0) continue;
    if (ix[k] >= v->map->N) { VecRestoreArray_HIP(v, NULL); SETERRQ(HipObjComm(v), PETSC_ERR_ARG_OUTOFRANGE, "Out of range index value %d maximum %d", ix[k], v->map->N); }
    if (ix[k] < v->map->rstart || ix[k] >= v->map->rend) {   /* VecSetValues_MPI (pdvec.c): stashed until VecAssemblyBegin */
      ierr = HipStashAdd(&VH(v)->stash, ix[k], 0, y[k], (int)mode);
      if (ierr) { VecRestoreArray_HIP(v, NULL); CHKERRQ(ierr); }
      continue;
    }
    if (mode == INSERT_VALUES) a[ix[k] - v->map->rstart] = y[k];
    else a[ix[k] - v->map->rstart] += y[k];
  }
  return VecRestoreArray_HIP(v, NULL);
}

/* VecAssemblyBegin_MPI / End_MPI (pdvec.c): the stashed off-process entries reach their owners, rank after rank */
static PetscErrorCode VecAssemblyBegin_HIP(Vec v) {
  PetscErrorCode ierr;
  PetscInt nr, *ri, *rj; PetscScalar *rv, *a = NULL; int smode;
  ierr = HipStashExchange(HipObjComm(v), &VH(v)->stash, &nr, &ri, &rj, &rv, &smode);CHKERRQ(ierr);
  if (nr) {
    ierr = VecGetArray_HIP(v, &a);CHKERRQ(ierr);
    for (PetscInt k = 0; k < nr; k++) {
      if (ri[k] < v->map->rstart || ri[k] >= v->map->rend) continue;
      if (smode == (int)INSERT_VALUES) a[ri[k] - v->map->rstart] = rv[k];
      else a[ri[k] - v->map->rstart] += rv[k];
    }
    ierr = VecRestoreArray_HIP(v, NULL);CHKERRQ(ierr);
    HipStateIncrease(v);
  }
  HipFree(ri); HipFree(rj); HipFree(rv);
  return 0;
}

/* ---- element-wise ops ---- */
#define DEVCTX PetscDeviceCtx *dc; ierr = PetscDeviceGet(&dc);CHKERRQ(ierr)
#define N_(v) ((size_t)(v)->map->n)

static PetscErrorCode VecSet_HIP(Vec x, PetscScalar alpha) {
  PetscErrorCode ierr; PetscScalar *d; DEVCTX;
  ierr = VecHIPGetWrite(x, &d);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_set(dc->h, N_(x), alpha, d));
  return VecHIPRestoreWrite(x);
}
static PetscErrorCode VecCopy_HIP_now(Vec x, Vec y);
static PetscErrorCode VecCopy_HIP(Vec x, Vec y) {
  CheckHIP(y);
  /* PCApply_None (src/ksp/pc/impls/none/none.c: VecCopy(x, y)) as the third operation of the noted CG sweep: z = r */
  if (dq.n == 2 && !dq.busy && dq.op[0].kind == DQ_AXPY && x == dq.op[1].o && y != dq.op[0].o && y != dq.op[1].o && y != dq.op[0].a && is_hip(y) && y->map->n == x->map->n) {
    dq.op[2].kind = DQ_COPY; dq.op[2].o = y; dq.op[2].a = x; dq.op[2].b = NULL; dq.n = 3;
    return 0;
  }
  return VecCopy_HIP_now(x, y);
}
static PetscErrorCode VecCopy_HIP_now(Vec x, Vec y) {
  PetscErrorCode ierr; const PetscScalar *dx; PetscScalar *dy; DEVCTX;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(y, &dy);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_copy(dc->h, N_(x), dx, dy));
  return VecHIPRestoreWrite(y);
}
static PetscErrorCode VecSwap_HIP(Vec x, Vec y) {
  PetscErrorCode ierr; PetscScalar *dx, *dy; DEVCTX;
  CheckHIP(y);
  ierr = VecHIPGetReadWrite(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(y, &dy);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_swap(dc->h, N_(x), dx, dy));
  VecHIPRestoreWrite(x);
  return VecHIPRestoreWrite(y);
}
static PetscErrorCode VecScale_HIP(Vec x, PetscScalar alpha) {
  PetscErrorCode ierr; PetscScalar *d; DEVCTX;
  ierr = VecHIPGetReadWrite(x, &d);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_scale(dc->h, N_(x), alpha, d));
  ierr = PetscLogFlops((PetscLogDouble)x->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(x);
}
/* ---- deferred element-wise operations: the CG sweep of an UNCHANGED KSPSolve_CG ----------------------------------------------
 * PETSc's KSPSolve_CG (cg.c:206-232) updates with five calls -- VecAXPY(X,a,P); VecAXPY(R,-a,W); PCApply (Jacobi:
 * VecPointwiseMult(Z,R,D), jacobi.c:266-277); VecNorm(Z); VecTDot(Z,R) -- i.e. five kernels, 17 vector passes and two host waits
 * where the fused sweep (mi355x_vec_cg_update, the kernel of the registered cghipmi355x type) makes 8 passes and one wait with the
 * same bits.  An unchanged program never calls the fused entry point, so the TYPE recognises the sequence: a VecAXPY is not launched
 * at once but noted; a second one with the negated scalar on other vectors, then a VecPointwiseMult reading the second one's result,
 * extend the note; a VecNorm_2 / VecDot / VecTDot of the product then runs the whole note as ONE fused sweep, and the z'r it
 * leaves answers the VecTDot(Z,R) that follows without a kernel (kept per object state of z and r).  Anything else -- any access
 * to any vector's storage (VecHIPGetRead / Write, VecGetArray: every operation of this file and every Mat / PC / scatter routine
 * goes through them), a vector being destroyed, an operation that does not continue the pattern -- first runs what is noted, in
 * order, with the ordinary kernels: the same results as without the note.  -vec_hipmi355x_defer 0 switches it off. */
static int defer_on = -1;
static int defer_enabled(void) {
  if (defer_on < 0) { PetscInt v = 1; PetscBool set; if (PetscOptionsGetInt(NULL, "-vec_hipmi355x_defer", &v, &set)) v = 1; defer_on = v ? 1 : 0; }
  return defer_on;
}
static PetscErrorCode VecAXPY_HIP_now(Vec y, PetscScalar alpha, Vec x);
static PetscErrorCode VecPointwiseMult_HIP_now(Vec w, Vec x, Vec y);
static PetscErrorCode VecAXPBYPCZ_HIP_now(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y);
static PetscErrorCode VecWAXPY_HIP_now(Vec w, PetscScalar alpha, Vec x, Vec y);
static PetscErrorCode VecMAXPY_HIP_now(Vec y, PetscInt nv, const PetscScalar *alpha, Vec *x);
PetscErrorCode VecCGUpdate_HIPMI355X(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscScalar a, PetscScalar *zz, PetscScalar *zr, PetscScalar *rr, PetscBool *done);
PetscErrorCode VecPMultDot_HIPMI355X(Vec w, Vec x, Vec d, Vec y, PetscScalar *val, PetscBool *done);
PetscErrorCode VecPMultDotNorm2_HIPMI355X(Vec w, Vec x, Vec d, Vec s_, PetscScalar *dp, PetscReal *nm, PetscBool *done);
PetscErrorCode VecBCGSUpdate_HIPMI355X(Vec x, Vec r, Vec p, Vec s_, Vec t, Vec rp, PetscScalar alpha, PetscScalar omega, PetscScalar *rr, PetscScalar *rho, PetscBool *done);
static PetscErrorCode dq_run(const DqOp *o) {
  switch (o->kind) {
  case DQ_AXPY: return VecAXPY_HIP_now(o->o, o->s1, o->a);
  case DQ_PMULT: return VecPointwiseMult_HIP_now(o->o, o->a, o->b);
  case DQ_COPY: return VecCopy_HIP_now(o->a, o->o);
  case DQ_AXPBYPCZ: return VecAXPBYPCZ_HIP_now(o->o, o->s1, o->s2, o->s3, o->a, o->b);
  case DQ_WAXPY: return VecWAXPY_HIP_now(o->o, o->s1, o->a, o->b);
  case DQ_MAXPY: return VecMAXPY_HIP_now(o->o, dq.mnv, dq.malpha, dq.mx);
  default: return 0;
  }
}
/* ---- a noted product.  KSP_PCApplyBAorAB with left preconditioning is MatMult(A, x, T) followed by PCApply(T -> V) with a work
 * vector T nobody reads again (kspimpl.h; GMRES and BiCGStab do it once / twice per iteration).  MatMult of a sequential AIJ matrix is
 * therefore NOTED instead of launched; when PCApply_Jacobi's VecPointwiseMult(V, T, D) follows, V = D .* (A x) is one kernel with the
 * scaling in the product's epilogue (MatMultDiagonalScale, same bits) and T stays unwritten: any reader of T, any writer of x, a change
 * of A's device values, storage changing hands (VecShareArray) gets T = A x written first (MatMult's own kernel); a writer of T (the
 * next MatMult into the same work vector) simply makes it dead.  Anything else after the note runs the product as MatMult would have. */
static PetscErrorCode product_run(PendingProduct *q) {
  PetscErrorCode ierr;
  if (!q->on || pp_busy) return 0;
  const PendingProduct r = *q;
  q->on = 0; pp_busy = 1;
  ierr = (*r.now)(r.A, r.x, r.t);
  pp_busy = 0;
  CHKERRQ(ierr);
  return 0;
}
/* a vector is about to be written by something that does not pass the accessors' checks (pp_busy): the unwritten t is dead if it is
 * this vector, and is written first if its source is */
static PetscErrorCode product_before_write(Vec w) {
  if (!pl.on) return 0;
  if (w == pl.t) { pl.on = 0; return 0; }
  if (w == pl.x) return product_run(&pl);
  return 0;
}
PetscErrorCode VecHIPNoteProduct(Mat A, Vec x, Vec t, HipProductNowFn now, HipProductScaledFn scaled, PetscBool *noted) {
  PetscErrorCode ierr;
  *noted = PETSC_FALSE;
  if (!defer_enabled() || pp_busy || dq.busy || !is_hip(x) || !is_hip(t) || x == t || HipCommSize(HipObjComm(x)) > 1) return 0;
  FLUSH_DEFERRED();                                   /* noted operations (they may write x) and an older noted product run first */
  ierr = product_before_write(t);CHKERRQ(ierr);       /* this product will write t */
  pp.on = 1; pp.A = A; pp.astate = (long)HipObjState(A); pp.x = x; pp.t = t; pp.now = now; pp.scaled = scaled;
  *noted = PETSC_TRUE;
  return 0;
}
PetscErrorCode VecHIPProductMatrixChanges(Mat A) {    /* A's device values are about to change or go away */
  PetscErrorCode ierr;
  if (pp_busy) return 0;
  if (pp.on && pp.A == A) { ierr = product_run(&pp);CHKERRQ(ierr); }
  if (pl.on && pl.A == A) { ierr = product_run(&pl);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode deferred_flush(void) {
  PetscErrorCode ierr = 0;
  if (pp.on && !pp_busy) { ierr = product_run(&pp);CHKERRQ(ierr); }
  if (!dq.n || dq.busy) return 0;
  const int n = dq.n;
  DqOp ops[3];
  for (int k = 0; k < n; k++) ops[k] = dq.op[k];
  dq.n = 0; dq.busy = 1;
  for (int k = 0; k < n && !ierr; k++) ierr = dq_run(&ops[k]);
  dq.busy = 0;
  CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecHIPMI355XFlushDeferred(void) { PetscErrorCode ierr = deferred_flush();CHKERRQ(ierr); return product_run(&pl); }
/* on: 1 / 0; negative: as the options database says (-vec_hipmi355x_defer, read again at the next operation) */
PetscErrorCode VecHIPMI355XSetDeferral(PetscInt on) {
  PetscErrorCode ierr = deferred_flush();CHKERRQ(ierr);
  ierr = product_run(&pl);CHKERRQ(ierr);
  defer_on = on < 0 ? -1 : (on ? 1 : 0);
  dq.have = 0; memset(dq.la, 0, sizeof(dq.la)); memset(dq.lb, 0, sizeof(dq.lb));
  return 0;
}
static void dq_keep(Vec a, Vec b, PetscScalar v) { dq.ca = a; dq.cb = b; dq.sa = (long)HipObjState(a); dq.sb = (long)HipObjState(b); dq.cval = v; dq.have = 1; }
static int dq_kept(Vec x, Vec y, PetscScalar *v) {
  if (!dq.have || !((x == dq.ca && y == dq.cb) || (x == dq.cb && y == dq.ca))) return 0;
  if ((long)HipObjState(dq.ca) != dq.sa || (long)HipObjState(dq.cb) != dq.sb) return 0;
  *v = dq.cval;
  return 1;
}
/* the note holds the whole CG sweep: run it fused.  *zz, *zr = z'z, z'r (over all ranks); *done = PETSC_FALSE: not possible with these
 * vectors (then the note has been run with the ordinary kernels and the caller goes on as usual) */
static int dq_is_cg_sweep(void) { return dq.n == 3 && dq.op[0].kind == DQ_AXPY; }
static PetscErrorCode deferred_sweep(PetscScalar *zz, PetscScalar *zr, PetscBool *done) {
  PetscErrorCode ierr;
  PetscScalar rr;
  Vec x = dq.op[0].o, p = dq.op[0].a, r = dq.op[1].o, w = dq.op[1].a, z = dq.op[2].o;
  Vec d = dq.op[2].kind == DQ_COPY ? NULL : (dq.op[2].a == r ? dq.op[2].b : dq.op[2].a);       /* NULL: the copy of PCNONE */
  const PetscScalar a = dq.op[0].s1;
  *done = PETSC_FALSE;
  dq.busy = 1;                                        /* the accessors inside must not run the note */
  ierr = VecCGUpdate_HIPMI355X(x, r, z, p, w, d, a, zz, zr, &rr, done);
  dq.busy = 0;
  CHKERRQ(ierr);
  if (!*done) return deferred_flush();
  dq.n = 0;
  dq_keep(z, r, *zr);
  return 0;
}
/* BiCGStab (bcgs.c:100-125): VecAXPBYPCZ(X, alpha, omega, 1, P, S); VecWAXPY(R, -omega, T, S) noted, and VecNorm(R) or VecDot(R, RP)
 * asks: x += alpha p + omega s, r = s - omega t, r'r and (r, rp) in one sweep; rp: the other operand of that VecDot, or -- when the
 * norm asks first -- of the last VecDot this vector r was in (the (r, rp) of the previous iteration); (r, rp) is kept for the VecDot */
static int dq_is_bcgs_update(void) { return dq.n == 2 && dq.op[0].kind == DQ_AXPBYPCZ && dq.op[1].kind == DQ_WAXPY; }
static PetscErrorCode deferred_bcgs_update(Vec rp, PetscScalar *rr, PetscScalar *rho, PetscBool *done) {
  PetscErrorCode ierr;
  Vec x = dq.op[0].o, p = dq.op[0].a, s_ = dq.op[0].b, r = dq.op[1].o, t = dq.op[1].a;
  const PetscScalar alpha = dq.op[0].s1, omega = dq.op[0].s2;
  *done = PETSC_FALSE;
  if (!rp || !is_hip(rp)) return deferred_flush();
  dq.busy = 1;
  ierr = VecBCGSUpdate_HIPMI355X(x, r, p, s_, t, rp, alpha, omega, rr, rho, done);
  dq.busy = 0;
  CHKERRQ(ierr);
  if (!*done) return deferred_flush();
  dq.n = 0;
  dq_keep(r, rp, *rho);
  return 0;
}
static PetscErrorCode VecAXPY_HIP(Vec y, PetscScalar alpha, Vec x) {
  CheckHIP(x);
  if (alpha == 0.0) return 0;
  if (defer_enabled() && !dq.busy && x != y && is_hip(y) && x->map->n == y->map->n) {
    if (dq.n == 1 && dq.op[0].kind == DQ_AXPY && alpha == -dq.op[0].s1 && y != dq.op[0].o && y != dq.op[0].a && x != dq.op[0].o && x->map->n == dq.op[0].o->map->n) {
      dq.op[1].kind = DQ_AXPY; dq.op[1].o = y; dq.op[1].a = x; dq.op[1].b = NULL; dq.op[1].s1 = alpha; dq.n = 2;
      return 0;
    }
    FLUSH_DEFERRED();
    dq.op[0].kind = DQ_AXPY; dq.op[0].o = y; dq.op[0].a = x; dq.op[0].b = NULL; dq.op[0].s1 = alpha; dq.n = 1;
    return 0;
  }
  return VecAXPY_HIP_now(y, alpha, x);
}
static PetscErrorCode VecAXPY_HIP_now(Vec y, PetscScalar alpha, Vec x) {
  PetscErrorCode ierr; const PetscScalar *dx; PetscScalar *dy; DEVCTX;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(y, &dy);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_axpy(dc->h, N_(y), alpha, dx, dy));
  ierr = PetscLogFlops(2.0 * y->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(y);
}
static PetscErrorCode VecAYPX_HIP(Vec y, PetscScalar alpha, Vec x) {
  PetscErrorCode ierr; const PetscScalar *dx; PetscScalar *dy; DEVCTX;
  CheckHIP(x);
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(y, &dy);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_aypx(dc->h, N_(y), alpha, dx, dy));
  ierr = PetscLogFlops(2.0 * y->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(y);
}
static PetscErrorCode VecAXPBY_HIP(Vec y, PetscScalar alpha, PetscScalar beta, Vec x) {
  PetscErrorCode ierr; const PetscScalar *dx; PetscScalar *dy; DEVCTX;
  CheckHIP(x);
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(y, &dy);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_axpby(dc->h, N_(y), alpha, beta, dx, dy));
  ierr = PetscLogFlops(3.0 * y->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(y);
}
static PetscErrorCode VecWAXPY_HIP(Vec w, PetscScalar alpha, Vec x, Vec y) {
  CheckHIP(x); CheckHIP(y);
  /* BiCGStab's r = s - omega t behind the noted x = alpha p + omega s + x */
  if (dq.n == 1 && !dq.busy && dq.op[0].kind == DQ_AXPBYPCZ && alpha == -dq.op[0].s2 && y == dq.op[0].b && is_hip(w) && w != x && w != y &&
      w != dq.op[0].o && w != dq.op[0].a && x != dq.op[0].o && w->map->n == y->map->n && x->map->n == y->map->n) {
    dq.op[1].kind = DQ_WAXPY; dq.op[1].o = w; dq.op[1].a = x; dq.op[1].b = y; dq.op[1].s1 = alpha; dq.n = 2;
    return 0;
  }
  return VecWAXPY_HIP_now(w, alpha, x, y);
}
static PetscErrorCode VecWAXPY_HIP_now(Vec w, PetscScalar alpha, Vec x, Vec y) {
  PetscErrorCode ierr; const PetscScalar *dx, *dy; PetscScalar *dw; DEVCTX;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetRead(y, &dy);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(w, &dw);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_waxpy(dc->h, N_(w), alpha, dx, dy, dw));
  ierr = PetscLogFlops(2.0 * w->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(w);
}
static PetscErrorCode VecAXPBYPCZ_HIP(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y) {
  CheckHIP(x); CheckHIP(y);
  if (defer_enabled() && !dq.busy && gamma == 1.0 && is_hip(z) && z != x && z != y && x != y && x->map->n == z->map->n && y->map->n == z->map->n) {
    FLUSH_DEFERRED();                                /* BiCGStab's x = alpha p + omega s + x: the first operation of its update */
    dq.op[0].kind = DQ_AXPBYPCZ; dq.op[0].o = z; dq.op[0].a = x; dq.op[0].b = y; dq.op[0].s1 = alpha; dq.op[0].s2 = beta; dq.op[0].s3 = gamma; dq.n = 1;
    return 0;
  }
  return VecAXPBYPCZ_HIP_now(z, alpha, beta, gamma, x, y);
}
static PetscErrorCode VecAXPBYPCZ_HIP_now(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y) {
  PetscErrorCode ierr; const PetscScalar *dx, *dy; PetscScalar *dz; DEVCTX;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetRead(y, &dy);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(z, &dz);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_axpbypcz(dc->h, N_(z), alpha, beta, gamma, dx, dy, dz));
  ierr = PetscLogFlops(5.0 * z->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(z);
}
static PetscErrorCode VecPointwiseMult_HIP(Vec w, Vec x, Vec y) {
  CheckHIP(x); CheckHIP(y);
  if (pp.on && !pp_busy && !dq.busy && is_hip(w) && (x == pp.t) != (y == pp.t)) {   /* PCApply_Jacobi right behind a noted MatMult */
    Vec d = x == pp.t ? y : x;
    if (w != pp.t && w != pp.x && w != d && d != pp.x && pp.scaled && (long)HipObjState(pp.A) == pp.astate) {   /* (the matrix has not been touched since) */
      PetscBool ok = PETSC_FALSE;
      PetscErrorCode ierr = product_before_write(w);CHKERRQ(ierr);      /* an older unwritten work vector: dead if it is w */
      if (pl.on && d == pl.t) { ierr = product_run(&pl);CHKERRQ(ierr); }   /* (read as the diagonal: never in practice) */
      pp_busy = 1;
      ierr = (*pp.scaled)(pp.A, d, pp.x, w, &ok);
      pp_busy = 0;
      CHKERRQ(ierr);
      if (ok) {
        if (pl.on) { ierr = product_run(&pl);CHKERRQ(ierr); }          /* one unwritten work vector at a time */
        pl = pp; pp.on = 0;
        return 0;
      }
    }
  }
  if (dq.n == 2 && !dq.busy && dq.op[0].kind == DQ_AXPY && is_hip(w) && (x == dq.op[1].o) != (y == dq.op[1].o)) {   /* the CG sweep's z = r .* d */
    Vec r = dq.op[1].o, d = x == r ? y : x;
    if (w != dq.op[0].o && w != r && w != dq.op[0].a && w != d && d != dq.op[0].o && d != r && w->map->n == r->map->n && d->map->n == r->map->n) {
      dq.op[2].kind = DQ_PMULT; dq.op[2].o = w; dq.op[2].a = x; dq.op[2].b = y; dq.n = 3;
      return 0;
    }
  }
  if (defer_enabled() && !dq.busy && is_hip(w) && w != x && w != y && x->map->n == w->map->n && y->map->n == w->map->n) {
    FLUSH_DEFERRED();                                /* a product by itself (PCApply_Jacobi): a VecDot / VecDotNorm2 of it may follow (BiCGStab) */
    dq.op[0].kind = DQ_PMULT; dq.op[0].o = w; dq.op[0].a = x; dq.op[0].b = y; dq.n = 1;
    return 0;
  }
  return VecPointwiseMult_HIP_now(w, x, y);
}
static PetscErrorCode VecPointwiseMult_HIP_now(Vec w, Vec x, Vec y) {
  PetscErrorCode ierr; const PetscScalar *dx, *dy; PetscScalar *dw; DEVCTX;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetRead(y, &dy);CHKERRQ(ierr);
  if (w == x || w == y) { ierr = VecHIPGetReadWrite(w, &dw);CHKERRQ(ierr); }
  else { ierr = VecHIPGetWrite(w, &dw);CHKERRQ(ierr); }
  CHKHIP(mi355x_vec_pointwise_mult(dc->h, N_(w), dx, dy, dw));
  ierr = PetscLogFlops((PetscLogDouble)w->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(w);
}
static PetscErrorCode VecPointwiseDivide_HIP(Vec w, Vec x, Vec y) {
  PetscErrorCode ierr; const PetscScalar *dx, *dy; PetscScalar *dw; DEVCTX;
  CheckHIP(x); CheckHIP(y);
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetRead(y, &dy);CHKERRQ(ierr);
  if (w == x || w == y) { ierr = VecHIPGetReadWrite(w, &dw);CHKERRQ(ierr); }
  else { ierr = VecHIPGetWrite(w, &dw);CHKERRQ(ierr); }
  CHKHIP(mi355x_vec_pointwise_divide(dc->h, N_(w), dx, dy, dw));
  ierr = PetscLogFlops((PetscLogDouble)w->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(w);
}
static PetscErrorCode VecReciprocal_HIP(Vec x) {
  PetscErrorCode ierr; PetscScalar *d; DEVCTX;
  ierr = VecHIPGetReadWrite(x, &d);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_reciprocal(dc->h, N_(x), d));
  return VecHIPRestoreWrite(x);
}
static PetscErrorCode VecMAXPY_HIP(Vec y, PetscInt nv, const PetscScalar *alpha, Vec *x) {
  /* KSPGMRESClassicalGramSchmidtOrthogonalization's VecMAXPY(w, -h) (borthog2.c:60-66) is followed by VecNormalize = VecNorm + VecScale
   * (gmres.c:146): noted, it runs inside that VecNorm -- one sweep for the update and the sum of squares, same bits */
  if (defer_enabled() && !dq.busy && !pp_busy && nv >= 1 && nv <= 32 && is_hip(y)) {
    PetscBool ok = PETSC_TRUE;
    for (PetscInt j = 0; j < nv && ok; j++) ok = (PetscBool)(is_hip(x[j]) && x[j] != y && x[j]->map->n == y->map->n);
    if (ok) {
      FLUSH_DEFERRED();
      dq.op[0].kind = DQ_MAXPY; dq.op[0].o = y; dq.op[0].a = dq.op[0].b = NULL; dq.mnv = nv;
      for (PetscInt j = 0; j < nv; j++) { dq.malpha[j] = alpha[j]; dq.mx[j] = x[j]; }
      dq.n = 1;
      return 0;
    }
  }
  return VecMAXPY_HIP_now(y, nv, alpha, x);
}
static PetscErrorCode VecMAXPY_HIP_now(Vec y, PetscInt nv, const PetscScalar *alpha, Vec *x) {
  PetscErrorCode ierr; PetscScalar *dy; DEVCTX;
  const double **tab;
  ierr = PetscMalloc(sizeof(double *) * (size_t)nv, &tab);CHKERRQ(ierr);
  for (PetscInt j = 0; j < nv; j++) { CheckHIP(x[j]); ierr = VecHIPGetRead(x[j], &tab[j]);CHKERRQ(ierr); }
  ierr = VecHIPGetReadWrite(y, &dy);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_maxpy(dc->h, N_(y), nv, alpha, tab, dy));
  HipFree(tab);
  ierr = PetscLogFlops(nv * 2.0 * y->map->n);CHKERRQ(ierr);
  return VecHIPRestoreWrite(y);
}

/* ---- reductions ---- */
/* finish: `count` partial results are being written by the kernel just launched.  One rank: they
 * land in the pinned scratch.  Several ranks: they land in HBM scratch, are all-reduced in place
 * over RCCL on the same stream, copied to the pinned scratch, and only then do we synchronise. */
/* A communicator with an RCCL communicator attached reduces on the device (also a one-rank one: the tests use that
 * to run this path on a single GPU); several ranks without one use the host-staged transport. */
static int hip_local_only = 0;   /* ops->dot_local / norm_local / mdot_local: this rank's part only (comb.c:402-721) */
#define DEVICE_COLLECTIVES(x) (!hip_local_only && HipCommDevice(HipObjComm(x)) != NULL)
#define HOST_STAGED(x) (!hip_local_only && HipCommSize(HipObjComm(x)) > 1 && !HipCommDevice(HipObjComm(x)))
static PetscErrorCode reduce_target(Vec x, PetscDeviceCtx *dc, double **out) {
  *out = DEVICE_COLLECTIVES(x) ? mi355x_handle_device_scratch(dc->h) : mi355x_handle_host_scratch(dc->h);
  return 0;
}
/* count values were reduced on this rank into the target; the first nsum of them are summed (or max-ed) over the
 * ranks, the rest are carried along as they are */
static PetscErrorCode reduce_finish2(Vec x, PetscDeviceCtx *dc, int nsum, int count, int is_max, PetscScalar *result) {
  double *hs = mi355x_handle_host_scratch(dc->h);
  if (HOST_STAGED(x)) {
    /* host-staged transport (no RCCL communicator attached): the reference's own arrangement, a device
     * reduction followed by a host all-reduce (mpicusp.cu:32-113); used by the shared-GPU rehearsal tests */
    CHKHIP(mi355x_handle_synchronize(dc->h));
    { PetscErrorCode e_ = HipTriWatchCheck();CHKERRQ(e_); }
    for (int j = 0; j < count; j++) result[j] = hs[j];
    if (HipCommAllreduce(HipObjComm(x), result, nsum, 1, is_max ? 1 : 0)) SETERRQ(HipObjComm(x), PETSC_ERR_LIB, "allreduce failed");
    return 0;
  }
  if (DEVICE_COLLECTIVES(x)) {
    double *ds = mi355x_handle_device_scratch(dc->h);
    if (is_max) CHKHIP(mi355x_comm_allreduce_max(HipCommDevice(HipObjComm(x)), dc->h, ds, (size_t)nsum));
    else CHKHIP(mi355x_comm_allreduce_sum(HipCommDevice(HipObjComm(x)), dc->h, ds, (size_t)nsum));
    /* device -> pinned host by a tiny kernel on the same stream that also stores the completion number the host
     * polls: no stream synchronisation, and kernels queued behind it do not delay the result */
    CHKHIP(mi355x_handle_publish(dc->h, ds, count));
    CHKHIP(mi355x_handle_wait_result(dc->h)); { PetscErrorCode e_ = HipTriWatchCheck();CHKERRQ(e_); }
  } else {
    /* one rank: the reduction kernel wrote the result and then a completion number to pinned memory; polling that
     * word is cheaper than a stream synchronisation and lets the next launches go out at once */
    CHKHIP(mi355x_handle_wait_result(dc->h)); { PetscErrorCode e_ = HipTriWatchCheck();CHKERRQ(e_); }
  }
  for (int j = 0; j < count; j++) result[j] = hs[j];
  return 0;
}
static PetscErrorCode reduce_finish(Vec x, PetscDeviceCtx *dc, int count, int is_max, PetscScalar *result) {
  return reduce_finish2(x, dc, count, count, is_max, result);
}

static PetscErrorCode VecDot_HIP(Vec x, Vec y, PetscScalar *val) {
  PetscErrorCode ierr; const PetscScalar *dx, *dy; double *out; DEVCTX;
  CheckHIP(y);
  if (!hip_local_only && !dq.busy) {
    PetscBool done = PETSC_FALSE; PetscScalar u, v;
    if (x != y) { dq.la[dq.lpos & 3] = x; dq.lb[dq.lpos & 3] = y; dq.lpos++; }
    if (dq_is_cg_sweep() && ((x == dq.op[2].o && y == dq.op[1].o) || (x == dq.op[1].o && y == dq.op[2].o))) {   /* natural norm: the sweep's product is first asked for as z'r */
      ierr = deferred_sweep(&u, &v, &done);CHKERRQ(ierr);
      if (done) { *val = v; return 0; }
    } else if (dq_is_bcgs_update() && x != y && (x == dq.op[1].o || y == dq.op[1].o)) {                          /* BiCGStab without a norm: (r, rp) asks */
      ierr = deferred_bcgs_update(x == dq.op[1].o ? y : x, &u, &v, &done);CHKERRQ(ierr);
      if (done) { *val = v; return 0; }
    } else if (dq.n == 1 && dq.op[0].kind == DQ_PMULT && x != y && (x == dq.op[0].o || y == dq.op[0].o)) {       /* (K p, rp) right behind the Jacobi product */
      const DqOp o = dq.op[0];
      dq.busy = 1;
      ierr = VecPMultDot_HIPMI355X(o.o, o.a, o.b, x == o.o ? y : x, &v, &done);
      dq.busy = 0;
      CHKERRQ(ierr);
      if (done) { dq.n = 0; *val = v; return 0; }
    } else if (!dq.n && dq_kept(x, y, &v)) {          /* left by the fused sweep that answered the VecNorm before */
      *val = v;
      return 0;
    }
  }
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetRead(y, &dy);CHKERRQ(ierr);
  ierr = reduce_target(x, dc, &out);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_dot(dc->h, N_(x), dx, dy, out));
  ierr = reduce_finish(x, dc, 1, 0, val);CHKERRQ(ierr);
  if (x->map->n > 0) { ierr = PetscLogFlops(2.0 * x->map->n - 1);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode VecMDot_HIP(Vec x, PetscInt nv, const Vec y[], PetscScalar *val) {
  PetscErrorCode ierr; const PetscScalar *dx; double *out; DEVCTX;
  const double *tab[32];
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  for (PetscInt pos = 0; pos < nv; pos += 32) {   /* pvec2.c:12-17 uses a 128-entry stack buffer; ours is the 64-double scratch */
    PetscInt cnt = PetscMin(32, nv - pos);
    for (PetscInt j = 0; j < cnt; j++) { CheckHIP(y[pos + j]); ierr = VecHIPGetRead(y[pos + j], &tab[j]);CHKERRQ(ierr); }
    ierr = reduce_target(x, dc, &out);CHKERRQ(ierr);
    CHKHIP(mi355x_vec_mdot(dc->h, N_(x), cnt, dx, tab, out));
    ierr = reduce_finish(x, dc, cnt, 0, val + pos);CHKERRQ(ierr);
  }
  ierr = PetscLogFlops(PetscMax(nv * (2.0 * x->map->n - 1), 0.0));CHKERRQ(ierr);
  return 0;
}
/* the noted MAXPY and the sum of squares of its result in one sweep (mi355x_vec_maxpy_dev_norm2: VecMAXPY_Seq's grouping, VecNorm's order);
 * the coefficients go to the device through a pinned staging buffer, in stream order */
static PetscErrorCode deferred_maxpy_norm(PetscScalar *sumsq, PetscBool *done) {
  PetscErrorCode ierr; DEVCTX;
  static double *stage = NULL, *dcoef = NULL;
  const double *tab[32]; PetscScalar *dy; double *out;
  Vec y = dq.op[0].o; const PetscInt nv = dq.mnv;
  *done = PETSC_FALSE;
  if (!stage) {
    if (mi355x_host_malloc((void **)&stage, sizeof(double) * 32) || mi355x_malloc((void **)&dcoef, sizeof(double) * 32)) { stage = NULL; return deferred_flush(); }
  }
  dq.busy = 1;
  ierr = 0;
  for (PetscInt j = 0; j < nv && !ierr; j++) { stage[j] = dq.malpha[j]; ierr = VecHIPGetRead(dq.mx[j], &tab[j]); }
  if (!ierr) ierr = VecHIPGetReadWrite(y, &dy);
  dq.busy = 0;
  CHKERRQ(ierr);
  dq.n = 0;
  CHKHIP(mi355x_memcpy_h2d(dc->h, dcoef, stage, sizeof(double) * (size_t)nv));
  ierr = reduce_target(y, dc, &out);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_maxpy_dev_norm2(dc->h, N_(y), (int)nv, dcoef, 1.0, tab, dy, out));
  VecHIPRestoreWrite(y);
  HipStateIncrease(y);
  ierr = reduce_finish(y, dc, 1, 0, sumsq);CHKERRQ(ierr);
  ierr = PetscLogFlops(nv * 2.0 * y->map->n + PetscMax(2.0 * y->map->n - 1, 0.0));CHKERRQ(ierr);
  *done = PETSC_TRUE;
  return 0;
}
static PetscErrorCode VecNorm_HIP(Vec x, NormType type, PetscReal *val) {
  PetscErrorCode ierr; const PetscScalar *dx; double *out; PetscScalar r[2]; DEVCTX;
  if (!dq.busy && !hip_local_only && type == NORM_2) {
    PetscBool done = PETSC_FALSE; PetscScalar u, v;
    if (dq_is_cg_sweep() && x == dq.op[2].o) {
      ierr = deferred_sweep(&u, &v, &done);CHKERRQ(ierr);
      if (done) { *val = PetscSqrtReal(u); return 0; }
    } else if (dq.n == 1 && dq.op[0].kind == DQ_MAXPY && x == dq.op[0].o) {
      ierr = deferred_maxpy_norm(&u, &done);CHKERRQ(ierr);
      if (done) { *val = PetscSqrtReal(u); return 0; }
    } else if (dq_is_bcgs_update() && x == dq.op[1].o) {
      Vec rp = NULL;
      for (int k = 1; k <= 4 && !rp; k++) {           /* the newest product this r was in */
        const int q = (dq.lpos - k) & 3;
        if (dq.la[q] == x) rp = dq.lb[q]; else if (dq.lb[q] == x) rp = dq.la[q];
      }
      ierr = deferred_bcgs_update(rp, &u, &v, &done);CHKERRQ(ierr);
      if (done) { *val = PetscSqrtReal(u); return 0; }
    }
  }
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = reduce_target(x, dc, &out);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_norm(dc->h, N_(x), (int)type, dx, out));
  if (type == NORM_1_AND_2) {
    ierr = reduce_finish(x, dc, 2, 0, r);CHKERRQ(ierr);
    val[0] = r[0]; val[1] = PetscSqrtReal(r[1]);           /* pvec2.c:80-82 */
  } else if (type == NORM_INFINITY) {
    ierr = reduce_finish(x, dc, 1, 1, r);CHKERRQ(ierr);    /* MPIU_MAX, pvec2.c:74 */
    *val = r[0];
  } else {
    ierr = reduce_finish(x, dc, 1, 0, r);CHKERRQ(ierr);
    *val = (type == NORM_1) ? r[0] : PetscSqrtReal(r[0]);  /* pvec2.c:62-64: reduce the square, then sqrt */
  }
  if (type != NORM_INFINITY) { ierr = PetscLogFlops(PetscMax(2.0 * x->map->n - 1, 0.0));CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode VecDotNorm2_HIP(Vec s, Vec t, PetscScalar *dp, PetscScalar *nm) {
  PetscErrorCode ierr; const PetscScalar *ds_, *dt; double *out; PetscScalar r[2]; DEVCTX;
  CheckHIP(t);
  if (dq.n == 1 && !dq.busy && !hip_local_only && dq.op[0].kind == DQ_PMULT && t == dq.op[0].o && s != t) {   /* BiCGStab: (s, t), (t, t) right behind t = K s */
    const DqOp o = dq.op[0]; PetscBool done = PETSC_FALSE; PetscReal nrm2;
    dq.busy = 1;
    ierr = VecPMultDotNorm2_HIPMI355X(o.o, o.a, o.b, s, dp, &nrm2, &done);
    dq.busy = 0;
    CHKERRQ(ierr);
    if (done) { dq.n = 0; *nm = nrm2; return 0; }
  }
  ierr = VecHIPGetRead(s, &ds_);CHKERRQ(ierr);
  ierr = VecHIPGetRead(t, &dt);CHKERRQ(ierr);
  ierr = reduce_target(s, dc, &out);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_dotnorm2(dc->h, N_(s), ds_, dt, out));
  ierr = reduce_finish(s, dc, 2, 0, r);CHKERRQ(ierr);
  *dp = r[0]; *nm = r[1];
  ierr = PetscLogFlops(4.0 * s->map->n);CHKERRQ(ierr);
  return 0;
}

/* per-launch device timing of the fused CG update for bench.py (hipEvent pairs on the compute stream, like
 * MatHIPMI355XSetTiming for the product) */
static struct { PetscBool on; PetscInt n, cap; mi355x_event_t *ev; } cgu_time;
PetscErrorCode VecHIPMI355XSetCGUpdateTiming(PetscBool on) {
  cgu_time.on = on; cgu_time.n = 0;
  if (on && !cgu_time.ev) {
    cgu_time.cap = 4096;
    PetscErrorCode ierr = PetscMalloc(sizeof(mi355x_event_t) * 2 * (size_t)cgu_time.cap, &cgu_time.ev);CHKERRQ(ierr);
    for (PetscInt k = 0; k < 2 * cgu_time.cap; k++) CHKHIP(mi355x_event_create(&cgu_time.ev[k]));
  }
  return 0;
}
PetscErrorCode VecHIPMI355XGetCGUpdateTiming(PetscInt *nlaunches, PetscLogDouble *total_ms) {
  double tot = 0.0;
  for (PetscInt k = 0; k < cgu_time.n; k++) {
    float ms = 0.f;
    CHKHIP(mi355x_event_synchronize(cgu_time.ev[2 * k + 1]));
    CHKHIP(mi355x_event_elapsed_ms(cgu_time.ev[2 * k], cgu_time.ev[2 * k + 1], &ms));
    tot += ms;
  }
  if (nlaunches) *nlaunches = cgu_time.n;
  if (total_ms) *total_ms = tot;
  return 0;
}
#define CGU_TIME_BEGIN(h) do { if (cgu_time.on && cgu_time.n < cgu_time.cap) CHKHIP(mi355x_event_record(cgu_time.ev[2 * cgu_time.n], h)); } while (0)
#define CGU_TIME_END(h)   do { if (cgu_time.on && cgu_time.n < cgu_time.cap) { CHKHIP(mi355x_event_record(cgu_time.ev[2 * cgu_time.n + 1], h)); cgu_time.n++; } } while (0)

/* Fused CG update for HIPMI355X vectors (KSPSolve_CG cg.c:206-232 with a Jacobi PCApply in the middle):
 * x += a p; r -= a w; z = r .* d; *zz = z'z, *zr = z'r in one sweep and one reduction (one all-reduce of two
 * doubles on several ranks instead of two of one).  Results carry the bits of the separate VecAXPY, VecAXPY,
 * VecPointwiseMult, VecNorm, VecTDot calls.  Returns *done = PETSC_FALSE (and does nothing) when an operand is
 * not a HIPMI355X vector, so the caller keeps the unfused sequence. */
PetscErrorCode VecCGUpdate_HIPMI355X(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscScalar a, PetscScalar *zz, PetscScalar *zr, PetscScalar *rr, PetscBool *done) {
  PetscErrorCode ierr; const PetscScalar *dp_, *dw, *dd; PetscScalar *dx, *dr, *dz; double *out; PetscScalar res[3]; DEVCTX;
  *done = PETSC_FALSE;
  ierr = VecCGUpdateCheck_HIPMI355X(x, r, z, p, w, d, done);CHKERRQ(ierr);
  if (!*done || a == 0.0) { *done = PETSC_FALSE; return 0; }
  *done = PETSC_FALSE;
  ierr = VecHIPGetRead(p, &dp_);CHKERRQ(ierr);
  ierr = VecHIPGetRead(w, &dw);CHKERRQ(ierr);
  dd = NULL;
  if (d) { ierr = VecHIPGetRead(d, &dd);CHKERRQ(ierr); }
  ierr = VecHIPGetReadWrite(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(r, &dr);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(z, &dz);CHKERRQ(ierr);
  ierr = reduce_target(x, dc, &out);CHKERRQ(ierr);
  CGU_TIME_BEGIN(dc->h);
  CHKHIP(mi355x_vec_cg_update(dc->h, N_(x), a, dp_, dw, dd, dx, dr, dz, out));
  CGU_TIME_END(dc->h);
  VecHIPRestoreWrite(x); VecHIPRestoreWrite(r); VecHIPRestoreWrite(z);
  HipStateIncrease(x); HipStateIncrease(r); HipStateIncrease(z);
  ierr = reduce_finish(x, dc, 3, 0, res);CHKERRQ(ierr);
  *zz = res[0]; *zr = res[1]; *rr = res[2];
  ierr = PetscLogFlops(9.0 * x->map->n);CHKERRQ(ierr);   /* 2n + 2n + n + 2n + 2n */
  *done = PETSC_TRUE;
  return 0;
}

/* Split form for KSPSolve_CG with device-resident scalars.  VecTDotBegin leaves p'w in slot DPI_SLOT of the
 * device scratch (all-reduced in place over RCCL when the communicator has one) without any host synchronisation;
 * VecCGUpdateDev then forms a = beta/dpi on the device, does the fused update and returns z'z, z'r and dpi with the
 * one synchronisation of the iteration.  Not available with the host-staged transport (*ok = PETSC_FALSE). */
#define DPI_SLOT PETSC_HIP_DPI_SLOT
PetscErrorCode VecTDotBegin_HIPMI355X(Vec x, Vec y, PetscBool *ok) {
  PetscErrorCode ierr; const PetscScalar *dx, *dy; double *ds; DEVCTX;
  *ok = PETSC_FALSE;
  if (!is_hip(x) || !is_hip(y) || HOST_STAGED(x) || x->map->n != y->map->n) return 0;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetRead(y, &dy);CHKERRQ(ierr);
  ds = mi355x_handle_device_scratch(dc->h) + DPI_SLOT;
  CHKHIP(mi355x_vec_dot(dc->h, N_(x), dx, dy, ds));
  if (DEVICE_COLLECTIVES(x)) CHKHIP(mi355x_comm_allreduce_sum(HipCommDevice(HipObjComm(x)), dc->h, ds, 1));
  if (x->map->n > 0) { ierr = PetscLogFlops(2.0 * x->map->n - 1);CHKERRQ(ierr); }
  *ok = PETSC_TRUE;
  return 0;
}
/* Begin queues the sweep (results go to device scratch slots 0..2, all-reduced there when the communicator has an
 * RCCL communicator, then published to pinned memory); End waits for exactly that point of the stream, so the caller
 * may queue more work in between (KSPSolve_CG queues the next iteration's AYPX, MatMult and dot there). */
PetscErrorCode VecCGUpdateDevBegin_HIPMI355X(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscScalar beta, PetscScalar dpiold, PetscBool check_sign) {
  PetscErrorCode ierr; const PetscScalar *dp_, *dw, *dd; PetscScalar *dx, *dr, *dz; double *ds; DEVCTX;
  ierr = VecHIPGetRead(p, &dp_);CHKERRQ(ierr);
  ierr = VecHIPGetRead(w, &dw);CHKERRQ(ierr);
  dd = NULL;
  if (d) { ierr = VecHIPGetRead(d, &dd);CHKERRQ(ierr); }
  ierr = VecHIPGetReadWrite(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(r, &dr);CHKERRQ(ierr);
  if (z == w) { ierr = VecHIPGetReadWrite(z, &dz);CHKERRQ(ierr); }   /* left as it is when the update is refused */
  else { ierr = VecHIPGetWrite(z, &dz);CHKERRQ(ierr); }
  ds = mi355x_handle_device_scratch(dc->h);
  /* one rank: the finishing workgroup hands the sums to the host itself; several: all-reduce first, then publish */
  CGU_TIME_BEGIN(dc->h);
  CHKHIP(mi355x_vec_cg_update_dev(dc->h, N_(x), beta, ds + DPI_SLOT, dpiold, (int)check_sign, dp_, dw, dd, dx, dr, dz, ds,
                                  DEVICE_COLLECTIVES(x) ? 0 : 1));
  CGU_TIME_END(dc->h);
  VecHIPRestoreWrite(x); VecHIPRestoreWrite(r); VecHIPRestoreWrite(z);
  HipStateIncrease(x); HipStateIncrease(r); HipStateIncrease(z);
  if (DEVICE_COLLECTIVES(x)) {
    CHKHIP(mi355x_comm_allreduce_sum(HipCommDevice(HipObjComm(x)), dc->h, ds, 3));
    CHKHIP(mi355x_handle_publish(dc->h, ds, 4));
  }
  ierr = PetscLogFlops(9.0 * x->map->n);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecCGUpdateDevEnd_HIPMI355X(Vec x, PetscScalar *zz, PetscScalar *zr, PetscScalar *rr, PetscScalar *dpi) {
  PetscErrorCode ierr; DEVCTX;
  (void)x;
  CHKHIP(mi355x_handle_wait_result(dc->h)); { PetscErrorCode e_ = HipTriWatchCheck();CHKERRQ(e_); }
  const double *hs = mi355x_handle_host_scratch(dc->h);
  *zz = hs[0]; *zr = hs[1]; *rr = hs[2]; *dpi = hs[3];
  return 0;
}
/* p = z + (zr/den) p with zr = the z'r the last VecCGUpdateDevBegin left in device scratch slot 1 (VecAYPX(P,b,Z) of
 * cg.c:187 with b = beta_new/beta_old formed on the device) */
PetscErrorCode VecAYPXDev_HIPMI355X(Vec p, PetscScalar den, Vec z) {
  PetscErrorCode ierr; const PetscScalar *dz; PetscScalar *dp_; DEVCTX;
  ierr = VecHIPGetRead(z, &dz);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(p, &dp_);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_aypx_dev(dc->h, N_(p), mi355x_handle_device_scratch(dc->h) + 1, den, dz, dp_));
  VecHIPRestoreWrite(p);
  HipStateIncrease(p);
  ierr = PetscLogFlops(2.0 * p->map->n);CHKERRQ(ierr);
  return 0;
}
/* can the six vectors of a CG iteration take the fused update at all (types, sizes, aliasing)? */
PetscErrorCode VecCGUpdateCheck_HIPMI355X(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscBool *ok) {
  *ok = PETSC_FALSE;
  /* d == NULL: identity preconditioner (PCNONE), z = r */
  if (!is_hip(x) || !is_hip(r) || !is_hip(z) || !is_hip(p) || !is_hip(w) || (d && !is_hip(d))) return 0;
  if (x->map->n != r->map->n || x->map->n != z->map->n || x->map->n != p->map->n || x->map->n != w->map->n || (d && x->map->n != d->map->n)) return 0;
  if (x == r || x == z || r == z || z == p || z == d) return 0;   /* z may be w (cg.c:122 keeps A*p in Z) */
  *ok = PETSC_TRUE;
  return 0;
}

/* Fused forms for KSPSolve_BCGS (krylov.c); *done = PETSC_FALSE and nothing touched unless every operand is a
 * HIPMI355X vector of the same local size.  d == NULL stands for the identity preconditioner. */
static int all_hip_same_size(Vec a, Vec b, Vec c, Vec d, Vec e, Vec f) {
  Vec v[6] = {a, b, c, d, e, f};
  for (int i = 0; i < 6; i++) if (v[i] && (!is_hip(v[i]) || v[i]->map->n != a->map->n)) return 0;
  return 1;
}
PetscErrorCode VecPMultDot_HIPMI355X(Vec w, Vec x, Vec d, Vec y, PetscScalar *val, PetscBool *done) {   /* w = x.*d ; val = (w,y) */
  PetscErrorCode ierr; const PetscScalar *dx, *dd = NULL, *dy; PetscScalar *dw; double *out; DEVCTX;
  *done = PETSC_FALSE;
  if (!all_hip_same_size(w, x, y, d, NULL, NULL) || w == x || w == y || w == d) return 0;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  if (d) { ierr = VecHIPGetRead(d, &dd);CHKERRQ(ierr); }
  ierr = VecHIPGetRead(y, &dy);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(w, &dw);CHKERRQ(ierr);
  ierr = reduce_target(w, dc, &out);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_pmult_dot(dc->h, N_(w), dx, dd, dy, dw, out));
  VecHIPRestoreWrite(w); HipStateIncrease(w);
  ierr = reduce_finish(w, dc, 1, 0, val);CHKERRQ(ierr);
  ierr = PetscLogFlops(3.0 * w->map->n);CHKERRQ(ierr);
  *done = PETSC_TRUE;
  return 0;
}
PetscErrorCode VecPMultDotNorm2_HIPMI355X(Vec w, Vec x, Vec d, Vec s_, PetscScalar *dp, PetscReal *nm, PetscBool *done) {   /* w = x.*d ; (s,w), (w,w) */
  PetscErrorCode ierr; const PetscScalar *dx, *dd = NULL, *dsv; PetscScalar *dw; double *out; PetscScalar res[2]; DEVCTX;
  *done = PETSC_FALSE;
  if (!all_hip_same_size(w, x, s_, d, NULL, NULL) || w == x || w == s_ || w == d) return 0;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  if (d) { ierr = VecHIPGetRead(d, &dd);CHKERRQ(ierr); }
  ierr = VecHIPGetRead(s_, &dsv);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(w, &dw);CHKERRQ(ierr);
  ierr = reduce_target(w, dc, &out);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_pmult_dotnorm2(dc->h, N_(w), dx, dd, dsv, dw, out));
  VecHIPRestoreWrite(w); HipStateIncrease(w);
  ierr = reduce_finish(w, dc, 2, 0, res);CHKERRQ(ierr);
  *dp = res[0]; *nm = res[1];
  ierr = PetscLogFlops(5.0 * w->map->n);CHKERRQ(ierr);
  *done = PETSC_TRUE;
  return 0;
}
/* x = alpha p + omega s + x ; r = s - omega t ; *rr = (r,r) ; *rho = (r,rp) */
PetscErrorCode VecBCGSUpdate_HIPMI355X(Vec x, Vec r, Vec p, Vec s_, Vec t, Vec rp, PetscScalar alpha, PetscScalar omega, PetscScalar *rr, PetscScalar *rho, PetscBool *done) {
  PetscErrorCode ierr; const PetscScalar *dp_, *dsv, *dt, *drp; PetscScalar *dx, *dr; double *out; PetscScalar res[2]; DEVCTX;
  *done = PETSC_FALSE;
  if (!all_hip_same_size(x, r, p, s_, t, rp) || x == r || x == p || x == s_ || x == t || x == rp || r == p || r == s_ || r == t || r == rp) return 0;
  ierr = VecHIPGetRead(p, &dp_);CHKERRQ(ierr);
  ierr = VecHIPGetRead(s_, &dsv);CHKERRQ(ierr);
  ierr = VecHIPGetRead(t, &dt);CHKERRQ(ierr);
  ierr = VecHIPGetRead(rp, &drp);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(r, &dr);CHKERRQ(ierr);
  ierr = reduce_target(x, dc, &out);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_bcgs_update(dc->h, N_(x), alpha, omega, dp_, dsv, dt, drp, dx, dr, out));
  VecHIPRestoreWrite(x); VecHIPRestoreWrite(r);
  HipStateIncrease(x); HipStateIncrease(r);
  ierr = reduce_finish(x, dc, 2, 0, res);CHKERRQ(ierr);
  *rr = res[0]; *rho = res[1];
  ierr = PetscLogFlops(10.0 * x->map->n);CHKERRQ(ierr);
  *done = PETSC_TRUE;
  return 0;
}

/* ---- split-phase reductions (src/vec/vec/utils/comb.c:402-721: VecDotBegin/End, VecNormBegin/End,
 * PetscCommSplitReductionBegin).  Begin launches the device reduction into the next slot of the device scratch (no
 * host involvement); PetscCommSplitReductionBegin starts what brings the queued results to the host -- one RCCL
 * all-reduce of all slots ON THE HALO STREAM (it overlaps the kernels the caller queues on the compute stream
 * meanwhile, the point of KSPGROPPCG) followed by the publish kernel there; on one rank just the publish kernel on the
 * compute stream, ahead of the caller's next kernels -- and the first End polls the completion number. */
#define SR_SLOT0 48      /* device scratch slots of the pending results (ordinary reductions use <= 32 from slot 0) */
#define SR_HOST0 48      /* where they land in the pinned scratch: clear of what ordinary reductions (<= 32 values) use meanwhile */
#define SR_MAX 8
static struct { int n, started, fetched; int kind[SR_MAX]; PetscScalar val[SR_MAX]; Vec owner; mi355x_event_t ev; } sr;
static PetscErrorCode sr_queue(Vec x, int kind, const PetscScalar *dx, const PetscScalar *dy) {
  PetscErrorCode ierr; DEVCTX;
  if (sr.started) SETERRQ(HipObjComm(x), PETSC_ERR_ORDER, "Called VecxxxBegin() in a different order or number of times than VecxxxEnd(): a split reduction is still pending");
  if (sr.n >= SR_MAX) SETERRQ(HipObjComm(x), PETSC_ERR_SUP, "at most %d reductions per split phase", SR_MAX);
  double *slot = HOST_STAGED(x) ? mi355x_handle_host_scratch(dc->h) + SR_HOST0 + sr.n : mi355x_handle_device_scratch(dc->h) + SR_SLOT0 + sr.n;
  if (kind == 0) CHKHIP(mi355x_vec_dot(dc->h, N_(x), dx, dy, slot));
  else CHKHIP(mi355x_vec_norm(dc->h, N_(x), 1, dx, slot));
  sr.kind[sr.n++] = kind; sr.owner = x;
  return 0;
}
static PetscErrorCode VecDotBegin_HIP(Vec x, Vec y, PetscScalar *result) {
  PetscErrorCode ierr; const PetscScalar *dx, *dy;
  (void)result;
  CheckHIP(x); CheckHIP(y);
  if (x->map->n != y->map->n) SETERRQ(HipObjComm(x), PETSC_ERR_ARG_INCOMP, "Incompatible vector local lengths");
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetRead(y, &dy);CHKERRQ(ierr);
  ierr = sr_queue(x, 0, dx, dy);CHKERRQ(ierr);
  if (x->map->n > 0) { ierr = PetscLogFlops(2.0 * x->map->n - 1);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode VecNormBegin_HIP(Vec x, NormType type, PetscReal *result) {
  PetscErrorCode ierr; const PetscScalar *dx;
  (void)result;
  CheckHIP(x);
  if (type != NORM_2) SETERRQ(HipObjComm(x), PETSC_ERR_SUP, "split-phase norms: NORM_2 only");
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = sr_queue(x, 1, dx, NULL);CHKERRQ(ierr);
  ierr = PetscLogFlops(PetscMax(2.0 * x->map->n - 1, 0.0));CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode PetscCommSplitReductionBegin_HIP(MPI_Comm comm) {
  PetscErrorCode ierr; DEVCTX;
  (void)comm;
  if (!sr.n || sr.started) return 0;
  Vec x = sr.owner;
  sr.started = 1; sr.fetched = 0;
  if (HOST_STAGED(x)) return 0;                                   /* host all-reduce at the first End */
  double *ds = mi355x_handle_device_scratch(dc->h) + SR_SLOT0;
  if (DEVICE_COLLECTIVES(x)) {
    if (!sr.ev) CHKHIP(mi355x_event_create(&sr.ev));
    CHKHIP(mi355x_event_record(sr.ev, dc->h));                    /* the reductions queued so far */
    CHKHIP(mi355x_handle_wait_event(dc->hcomm, sr.ev));
    CHKHIP(mi355x_comm_allreduce_sum(HipCommDeviceHalo(HipObjComm(x)), dc->hcomm, ds, (size_t)sr.n));   /* halo stream: its own communicator */
    CHKHIP(mi355x_handle_publish_at(dc->hcomm, ds, sr.n, SR_HOST0));   /* lands in the HALO handle's pinned scratch */
  } else {
    CHKHIP(mi355x_handle_publish_at(dc->h, ds, sr.n, SR_HOST0));
  }
  return 0;
}
static PetscErrorCode sr_fetch(Vec x) {
  PetscErrorCode ierr; DEVCTX;
  if (!sr.n) SETERRQ(HipObjComm(x), PETSC_ERR_ORDER, "VecxxxEnd() without a matching VecxxxBegin()");
  if (!sr.started) { ierr = PetscCommSplitReductionBegin(HipObjComm(x));CHKERRQ(ierr); }
  if (sr.fetched == 0) {
    Vec o = sr.owner;
    if (HOST_STAGED(o)) {
      CHKHIP(mi355x_handle_synchronize(dc->h));
      const double *hs = mi355x_handle_host_scratch(dc->h) + SR_HOST0;
      for (int j = 0; j < sr.n; j++) sr.val[j] = hs[j];
      if (HipCommAllreduce(HipObjComm(o), sr.val, sr.n, 1, 0)) SETERRQ(HipObjComm(o), PETSC_ERR_LIB, "allreduce failed");
    } else {
      mi355x_handle_t hh = DEVICE_COLLECTIVES(o) ? dc->hcomm : dc->h;
      CHKHIP(mi355x_handle_wait_result(hh)); { PetscErrorCode e_ = HipTriWatchCheck();CHKERRQ(e_); }
      const double *hs = mi355x_handle_host_scratch(hh) + SR_HOST0;
      for (int j = 0; j < sr.n; j++) sr.val[j] = hs[j];
      /* the slots may be rewritten by the next phase only after the all-reduce that read them has finished */
      if (DEVICE_COLLECTIVES(o)) { CHKHIP(mi355x_event_record(sr.ev, dc->hcomm)); CHKHIP(mi355x_handle_wait_event(dc->h, sr.ev)); }
    }
  }
  return 0;
}
static PetscErrorCode sr_take(Vec x, int kind, PetscScalar *out) {
  PetscErrorCode ierr = sr_fetch(x);CHKERRQ(ierr);
  if (sr.fetched >= sr.n || sr.kind[sr.fetched] != kind) SETERRQ(HipObjComm(x), PETSC_ERR_ORDER, "Called VecxxxEnd() in a different order or number of times than VecxxxBegin()");
  *out = sr.val[sr.fetched++];
  if (sr.fetched == sr.n) { sr.n = 0; sr.started = 0; sr.fetched = 0; sr.owner = NULL; }
  return 0;
}
static PetscErrorCode VecDotEnd_HIP(Vec x, Vec y, PetscScalar *result) { (void)y; return sr_take(x, 0, result); }
static PetscErrorCode VecNormEnd_HIP(Vec x, NormType type, PetscReal *result) {
  PetscScalar v;
  (void)type;
  PetscErrorCode ierr = sr_take(x, 1, &v);CHKERRQ(ierr);
  *result = PetscSqrtReal(v);                                     /* squares reduced, then rooted (comb.c / pvec2.c:62) */
  return 0;
}

/* the *_local slots of struct _VecOps (vecimpl.h:262-266): the same device reductions without the step across ranks;
 * NORM_2 returns the root of the local sum, as VecNorm_Seq does for VecNormBegin (comb.c squares it again) */
static PetscErrorCode VecDot_HIP_local(Vec x, Vec y, PetscScalar *val) { hip_local_only = 1; PetscErrorCode ierr = VecDot_HIP(x, y, val); hip_local_only = 0; return ierr; }
static PetscErrorCode VecMDot_HIP_local(Vec x, PetscInt nv, const Vec y[], PetscScalar *val) { hip_local_only = 1; PetscErrorCode ierr = VecMDot_HIP(x, nv, y, val); hip_local_only = 0; return ierr; }
static PetscErrorCode VecNorm_HIP_local(Vec x, NormType type, PetscReal *val) { hip_local_only = 1; PetscErrorCode ierr = VecNorm_HIP(x, type, val); hip_local_only = 0; return ierr; }

#if defined(PETSCHIPMI355X_WITH_PETSC)
#include "vechipmi355x_ctor.h"      /* integration/petsc-3.3/: the slots PETSc's real struct _VecOps needs beyond the numerical ones */
#endif

static PetscErrorCode VecDestroy_HIP(Vec v) {
  Vec_HIPMI355X *s = VH(v);
  if (!s) return 0;
  FLUSH_DEFERRED();                                  /* a pending operation may name this vector */
  if (pl.on && !pp_busy) { if (v == pl.t) pl.on = 0; else if (v == pl.x) { PetscErrorCode e__ = product_run(&pl);CHKERRQ(e__); } }
  if (dq.ca == v || dq.cb == v) dq.have = 0;
  for (int k = 0; k < 4; k++) if (dq.la[k] == v || dq.lb[k] == v) dq.la[k] = dq.lb[k] = NULL;
  if (s->placed_save) { s->host = s->placed_save; s->placed_save = NULL; }
  if (s->alias_save) { s->dev = s->alias_save; s->alias_save = NULL; }   /* never free storage borrowed from another vector */
  if (s->dev) mi355x_free(s->dev);
  if (s->host && s->host_owned) HipFree(s->host);
  HipStashFree(&s->stash);
  HipFree(s);
  v->data = NULL;
  return 0;
}

static PetscErrorCode VecDuplicate_HIP(Vec v, Vec *newv) {   /* VecDuplicate_SeqCUSP / _MPICUSP: same type, same layout */
  PetscErrorCode ierr;
  Vec w;
  ierr = VecCreate(HipObjComm(v), &w);CHKERRQ(ierr);
  ierr = PetscLayoutReference(v->map, &w->map);CHKERRQ(ierr);
  ierr = VecSetType(w, HipObjTypeName(v));CHKERRQ(ierr);
  *newv = w;
  return 0;
}

/* ---- methods beyond the function table, offered by name (PetscObjectComposeFunction) ---- */
/* "VecJacobiInvert_C": d = (d == 0) ? 1 : 1/d on the device (PCSetUp_Jacobi's host loop, jacobi.c:182-190) */
static PetscErrorCode VecJacobiInvert_HIP(Vec d) {
  PetscErrorCode ierr; PetscScalar *dd; PetscDeviceCtx *dc;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = VecHIPGetReadWrite(d, &dd);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_jacobi_invert(dc->h, (size_t)d->map->n, dd, NULL));
  return VecHIPRestoreWrite(d);
}
/* "VecShareArrayBegin_C" / "VecShareArrayEnd_C": sub borrows parent's DEVICE storage (PCApply_BJacobi_Singleblock's
 * VecPlaceArray of host arrays, bjacobi.c:738-761, without the host round trip) */
static PetscErrorCode VecShareSubArrayBegin_HIP(Vec sub, Vec parent, PetscInt offset, PetscBool write) {
  PetscErrorCode ierr;
  Vec_HIPMI355X *s = VH(sub);
  PetscScalar *dp;
  if (!parent->data || !strstr(HipObjTypeName(parent), "hipmi355x")) SETERRQ(HipObjComm(sub), PETSC_ERR_ARG_NOTSAMETYPE, "cannot share the storage of a %s vector", HipObjTypeName(parent));
  FLUSH_DEFERRED();
  if (pl.on && !pp_busy) { PetscErrorCode e__ = product_run(&pl);CHKERRQ(e__); }   /* storage changes hands: nothing stays unwritten across it */
  if (s->alias_save) SETERRQ(HipObjComm(sub), PETSC_ERR_ARG_WRONGSTATE, "vector already shares another vector's storage");
  if (offset < 0 || offset + sub->map->n > parent->map->n) SETERRQ(HipObjComm(sub), PETSC_ERR_ARG_SIZ, "%d entries from offset %d do not fit a vector of local size %d", sub->map->n, offset, parent->map->n);
  /* a block's output slice: the parent keeps what the other blocks have written (read-write access), unless the slice is all of it */
  if (write && offset == 0 && sub->map->n == parent->map->n) { ierr = VecHIPGetWrite(parent, &dp);CHKERRQ(ierr); }
  else if (write) { ierr = VecHIPGetReadWrite(parent, &dp);CHKERRQ(ierr); }
  else { const PetscScalar *cp; ierr = VecHIPGetRead(parent, &cp);CHKERRQ(ierr); dp = (PetscScalar *)cp; }
  ierr = VecHIPGetWrite(sub, &s->alias_save);CHKERRQ(ierr);      /* makes sure sub owns device storage to come back to */
  s->alias_valid = s->valid;
  s->dev = dp + offset; s->valid = VALID_DEVICE;
  HipStateIncrease(sub);
  return 0;
}
static PetscErrorCode VecShareSubArrayEnd_HIP(Vec sub, Vec parent, PetscInt offset, PetscBool write) {
  Vec_HIPMI355X *s = VH(sub);
  (void)offset;
  if (!s->alias_save) return 0;
  FLUSH_DEFERRED();                                  /* an operation pending on the borrowed storage runs before it goes back */
  if (pl.on && !pp_busy) { PetscErrorCode e__ = product_run(&pl);CHKERRQ(e__); }
  s->dev = s->alias_save; s->valid = s->alias_valid; s->alias_save = NULL;
  HipStateIncrease(sub);
  if (write) return VecHIPRestoreWrite(parent);
  return 0;
}
static PetscErrorCode VecShareArrayBegin_HIP(Vec sub, Vec parent, PetscBool write) {
  if (sub->map->n != parent->map->n) SETERRQ(HipObjComm(sub), PETSC_ERR_ARG_SIZ, "local sizes %d and %d differ", sub->map->n, parent->map->n);
  return VecShareSubArrayBegin_HIP(sub, parent, 0, write);
}
static PetscErrorCode VecShareArrayEnd_HIP(Vec sub, Vec parent, PetscBool write) { return VecShareSubArrayEnd_HIP(sub, parent, 0, write); }
/* KSPGMRESClassicalGramSchmidtOrthogonalization without refinement (borthog2.c:60-66) followed by gmres.c:146's VecNormalize,
 * with the scalars staying on the device: VecMDot leaves <w, V_j> in device scratch (all-reduced there over RCCL when the
 * communicator has one), one sweep does w -= sum_j h_j V_j AND sum w^2 (mi355x_vec_maxpy_dev_norm2: VecMAXPY's grouping, VecNorm's
 * order), a tiny kernel hands h and |w|^2 to the host, and w *= 1/|w| reads |w|^2 from device memory while the host already
 * has its numbers.  One host wait per Krylov step instead of three; 2 nv + 5 vector passes instead of 2 nv + 6. */
PetscErrorCode VecGMRESOrthogNormalize_HIPMI355X(Vec w, PetscInt nv, const Vec V[], PetscScalar *dots, PetscReal *nrm, PetscBool *done) {
  PetscErrorCode ierr; const double *tab[32]; PetscScalar *dw; double *ds; DEVCTX;
  *done = PETSC_FALSE;
  if (!is_hip(w) || nv < 1 || nv > 32 || HOST_STAGED(w)) return 0;
  for (PetscInt j = 0; j < nv; j++) if (!is_hip(V[j]) || V[j] == w || V[j]->map->n != w->map->n) return 0;
  for (PetscInt j = 0; j < nv; j++) { ierr = VecHIPGetRead(V[j], &tab[j]);CHKERRQ(ierr); }
  ierr = VecHIPGetReadWrite(w, &dw);CHKERRQ(ierr);
  ds = mi355x_handle_device_scratch(dc->h);
  CHKHIP(mi355x_vec_mdot(dc->h, N_(w), (int)nv, dw, tab, ds));
  if (DEVICE_COLLECTIVES(w)) CHKHIP(mi355x_comm_allreduce_sum(HipCommDevice(HipObjComm(w)), dc->h, ds, (size_t)nv));
  CHKHIP(mi355x_vec_maxpy_dev_norm2(dc->h, N_(w), (int)nv, ds, -1.0, tab, dw, ds + nv));
  if (DEVICE_COLLECTIVES(w)) CHKHIP(mi355x_comm_allreduce_sum(HipCommDevice(HipObjComm(w)), dc->h, ds + nv, 1));
  CHKHIP(mi355x_handle_publish(dc->h, ds, (int)nv + 1));
  CHKHIP(mi355x_vec_scale_rnorm_dev(dc->h, N_(w), ds + nv, dw));
  VecHIPRestoreWrite(w);
  HipStateIncrease(w);
  CHKHIP(mi355x_handle_wait_result(dc->h)); { PetscErrorCode e_ = HipTriWatchCheck();CHKERRQ(e_); }
  const double *hs = mi355x_handle_host_scratch(dc->h);
  for (PetscInt j = 0; j < nv; j++) dots[j] = hs[j];
  *nrm = PetscSqrtReal(hs[nv]);                                 /* pvec2.c:62-64: the square is reduced, then the root */
  ierr = PetscLogFlops(PetscMax(nv * (2.0 * w->map->n - 1), 0.0) + 2.0 * nv * w->map->n + PetscMax(2.0 * w->map->n - 1, 0.0) + w->map->n);CHKERRQ(ierr);
  *done = PETSC_TRUE;
  return 0;
}
/* "VecKrylovFusedOps_C": the fused sweeps of KSPSolve_CG / KSPSolve_BCGS, bit-identical to the calls they replace */
static const VecKrylovFusedOps *VecKrylovFusedOps_HIP(void) {
  static const VecKrylovFusedOps ops = {
    VecCGUpdate_HIPMI355X, VecCGUpdateCheck_HIPMI355X, VecTDotBegin_HIPMI355X, VecCGUpdateDevBegin_HIPMI355X, VecCGUpdateDevEnd_HIPMI355X,
    VecAYPXDev_HIPMI355X, VecPMultDot_HIPMI355X, VecPMultDotNorm2_HIPMI355X, VecBCGSUpdate_HIPMI355X, VecGMRESOrthogNormalize_HIPMI355X};
  return &ops;
}
/* "VecSplitReductionOps_C": VecDotBegin/End, VecNormBegin/End, PetscCommSplitReductionBegin (comb.c:402-721) with the
 * all-reduce on the halo stream */
typedef struct {
  PetscErrorCode (*dot_begin)(Vec, Vec, PetscScalar *);
  PetscErrorCode (*dot_end)(Vec, Vec, PetscScalar *);
  PetscErrorCode (*norm_begin)(Vec, NormType, PetscReal *);
  PetscErrorCode (*norm_end)(Vec, NormType, PetscReal *);
  PetscErrorCode (*comm_begin)(MPI_Comm);
} VecSplitReductionOps;
static const VecSplitReductionOps *VecSplitReductionOps_HIP(void) {
  static const VecSplitReductionOps ops = {VecDotBegin_HIP, VecDotEnd_HIP, VecNormBegin_HIP, VecNormEnd_HIP, PetscCommSplitReductionBegin_HIP};
  return &ops;
}

/* the common part of VecCreate_SeqCUSP (veccusp.cu:1905-1947) / VecCreate_MPICUSP (mpicusp.cu:179-230): the function-table
 * slots those constructors override, and petscnative = PETSC_FALSE so that VecGetArray / VecRestoreArray of user code come
 * through ops->getarray / restorearray (vecimpl.h:375-434) */
static PetscErrorCode VecCreate_HIP_common(Vec v, const char *tname) {
  PetscErrorCode ierr;
  Vec_HIPMI355X *s;
  (void)VecDot_HIP_local; (void)VecMDot_HIP_local; (void)VecNorm_HIP_local;
#if defined(PETSCHIPMI355X_WITH_PETSC)
  ierr = PetscMemzero(v->ops, sizeof(struct _VecOps));CHKERRQ(ierr);
  ierr = VecCreate_HIP_petsc33(v, (PetscBool)!strcmp(tname, VECMPIHIPMI355X));CHKERRQ(ierr);
#endif
  ierr = PetscMalloc(sizeof(*s), &s);CHKERRQ(ierr);
  memset(s, 0, sizeof(*s));
  v->data = s;
  v->petscnative = PETSC_FALSE;
  ierr = PetscObjectChangeTypeName((PetscObject)v, tname);CHKERRQ(ierr);
  v->ops->duplicate = VecDuplicate_HIP;
  v->ops->dot = VecDot_HIP;
  v->ops->tdot = VecDot_HIP;
  v->ops->mdot = VecMDot_HIP;
  v->ops->mtdot = VecMDot_HIP;
  v->ops->norm = VecNorm_HIP;
  v->ops->scale = VecScale_HIP;
  v->ops->copy = VecCopy_HIP;
  v->ops->set = VecSet_HIP;
  v->ops->swap = VecSwap_HIP;
  v->ops->axpy = VecAXPY_HIP;
  v->ops->axpby = VecAXPBY_HIP;
  v->ops->maxpy = VecMAXPY_HIP;
  v->ops->aypx = VecAYPX_HIP;
  v->ops->waxpy = VecWAXPY_HIP;
  v->ops->axpbypcz = VecAXPBYPCZ_HIP;
  v->ops->pointwisemult = VecPointwiseMult_HIP;
  v->ops->pointwisedivide = VecPointwiseDivide_HIP;
  v->ops->setvalues = VecSetValues_HIP;
  v->ops->assemblybegin = VecAssemblyBegin_HIP;
  v->ops->getarray = VecGetArray_HIP;
  v->ops->restorearray = VecRestoreArray_HIP;
  v->ops->placearray = VecPlaceArray_HIP;
  v->ops->replacearray = VecReplaceArray_HIP;
  v->ops->resetarray = VecResetArray_HIP;
  v->ops->destroy = VecDestroy_HIP;
  v->ops->reciprocal = VecReciprocal_HIP;
  v->ops->dotnorm2 = VecDotNorm2_HIP;
  ierr = PetscObjectComposeFunction((PetscObject)v, "VecJacobiInvert_C", "VecJacobiInvert_HIP", (PetscVoidFunction)VecJacobiInvert_HIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)v, "VecShareArrayBegin_C", "VecShareArrayBegin_HIP", (PetscVoidFunction)VecShareArrayBegin_HIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)v, "VecShareArrayEnd_C", "VecShareArrayEnd_HIP", (PetscVoidFunction)VecShareArrayEnd_HIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)v, "VecShareSubArrayBegin_C", "VecShareSubArrayBegin_HIP", (PetscVoidFunction)VecShareSubArrayBegin_HIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)v, "VecShareSubArrayEnd_C", "VecShareSubArrayEnd_HIP", (PetscVoidFunction)VecShareSubArrayEnd_HIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)v, "VecKrylovFusedOps_C", "VecKrylovFusedOps_HIP", (PetscVoidFunction)VecKrylovFusedOps_HIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)v, "VecSplitReductionOps_C", "VecSplitReductionOps_HIP", (PetscVoidFunction)VecSplitReductionOps_HIP);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode VecCreate_SeqHIPMI355X(Vec v) {
  if (HipCommSize(HipObjComm(v)) > 1) SETERRQ(HipObjComm(v), PETSC_ERR_ARG_WRONG, "Cannot create VECSEQHIPMI355X on more than one process");
  return VecCreate_HIP_common(v, VECSEQHIPMI355X);
}
PetscErrorCode VecCreate_MPIHIPMI355X(Vec v) { return VecCreate_HIP_common(v, VECMPIHIPMI355X); }
/* size dispatch, as VecCreate_CUSP mpicusp.cu:232-246 */
PetscErrorCode VecCreate_HIPMI355X(Vec v) {
  return (HipCommSize(HipObjComm(v)) == 1) ? VecCreate_SeqHIPMI355X(v) : VecCreate_MPIHIPMI355X(v);
}
PetscErrorCode VecCreateSeqHIPMI355X(MPI_Comm comm, PetscInt n, Vec *v) {
  PetscErrorCode ierr;
  ierr = VecCreate(comm, v);CHKERRQ(ierr);
  ierr = VecSetSizes(*v, n, n);CHKERRQ(ierr);
  ierr = VecSetType(*v, VECSEQHIPMI355X);CHKERRQ(ierr);
  return 0;
}

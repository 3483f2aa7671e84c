/* Vec: the public wrappers (argument checks, norm cache, state bumps -- src/vec/vec/interface/rvector.c)
 * and the type registry (src/vec/vec/interface/vecreg.c).  Every wrapper dispatches through v->ops. */
#include "petscimpl.h"

#define MAXTYPES 16
typedef PetscErrorCode (*VecCreateFn)(Vec);
static struct { char name[32]; VecCreateFn fn; } vec_types[MAXTYPES];
static int n_vec_types = 0;

PetscErrorCode VecRegister(const char name[], const char path[], const char fname[], VecCreateFn fn) {
  (void)path; (void)fname;
  for (int i = 0; i < n_vec_types; i++) if (!strcmp(vec_types[i].name, name)) { vec_types[i].fn = fn; return 0; }
  if (n_vec_types >= MAXTYPES) SETERRQ(0, PETSC_ERR_PLIB, "Vec type table full");
  snprintf(vec_types[n_vec_types].name, 32, "%s", name);
  vec_types[n_vec_types++].fn = fn;
  return 0;
}

#define VecValid(v, arg) do { if (!(v)) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null Object: Parameter # %d", arg); } while (0)
#define VecTypeSet(v, arg) do { VecValid(v, arg); if (!(v)->data) SETERRQ((v)->comm, PETSC_ERR_ARG_TYPENOTSET, "Vec type not set: Parameter # %d", arg); } while (0)
#define VecSameSize(x, y) do { \
    if ((x)->map->N != (y)->map->N) SETERRQ((x)->comm, PETSC_ERR_ARG_INCOMP, "Incompatible vector global lengths %d != %d", (x)->map->N, (y)->map->N); \
    if ((x)->map->n != (y)->map->n) SETERRQ((x)->comm, PETSC_ERR_ARG_INCOMP, "Incompatible vector local lengths %d != %d", (x)->map->n, (y)->map->n); } while (0)

PetscErrorCode VecCreate(PetscComm comm, Vec *vec) {
  PetscErrorCode ierr;
  Vec v;
  ierr = PetscMalloc(sizeof(*v), &v);CHKERRQ(ierr);
  memset(v, 0, sizeof(*v));
  v->comm = comm;
  for (int i = 0; i < 4; i++) v->norm_state[i] = -1;
  *vec = v;
  return 0;
}

PetscErrorCode VecSetSizes(Vec v, PetscInt n, PetscInt N) {
  PetscErrorCode ierr;
  VecValid(v, 1);
  if (N > 0 && n > N) SETERRQ(v->comm, PETSC_ERR_ARG_INCOMP, "Local size %d cannot be larger than global size %d", n, N);
  if (v->map) SETERRQ(v->comm, PETSC_ERR_SUP, "Cannot change/reset vector sizes");
  ierr = PetscLayoutCreateSetUp(v->comm, n, N, &v->map);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode VecSetType(Vec v, VecType type) {
  PetscErrorCode ierr;
  VecValid(v, 1);
  if (!strcmp(v->type_name, type)) return 0;
  if (!v->map) SETERRQ(v->comm, PETSC_ERR_ORDER, "Must call VecSetSizes() before VecSetType()");
  for (int i = 0; i < n_vec_types; i++) {
    if (!strcmp(vec_types[i].name, type)) {
      if (v->ops->destroy) { ierr = (*v->ops->destroy)(v);CHKERRQ(ierr); }
      memset(v->ops, 0, sizeof(v->ops));
      ierr = PetscObjectListDestroy_Private((PetscObject)v);CHKERRQ(ierr);
      v->data = NULL; v->petscnative = PETSC_FALSE;
      ierr = (*vec_types[i].fn)(v);CHKERRQ(ierr);
      return 0;
    }
  }
  SETERRQ(v->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown vector type: %s", type);
}

PetscErrorCode VecSetFromOptions(Vec v) {
  char t[64];
  PetscBool set;
  PetscErrorCode ierr = PetscOptionsGetString(NULL, "-vec_type", t, sizeof(t), &set);CHKERRQ(ierr);
  if (!set) snprintf(t, sizeof(t), "%s", VECSTANDARD);   /* vector.c VecSetFromOptions: standard = seq on one process, mpi on several */
  return VecSetType(v, t);
}
PetscErrorCode VecGetType(Vec v, VecType *type) { VecValid(v, 1); *type = v->type_name; return 0; }

PetscErrorCode VecDuplicate(Vec v, Vec *newv) {
  VecTypeSet(v, 1);
  PetscErrorCode ierr = (*v->ops->duplicate)(v, newv);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecDuplicateVecs(Vec v, PetscInt m, Vec **V) {
  PetscErrorCode ierr;
  VecTypeSet(v, 1);
  if (m <= 0) SETERRQ(v->comm, PETSC_ERR_ARG_OUTOFRANGE, "m must be > 0: m = %d", m);
  ierr = PetscMalloc(sizeof(Vec) * (size_t)m, V);CHKERRQ(ierr);
  for (PetscInt i = 0; i < m; i++) { ierr = VecDuplicate(v, &(*V)[i]);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode VecDestroyVecs(PetscInt m, Vec **V) {
  if (!*V) return 0;
  for (PetscInt i = 0; i < m; i++) { PetscErrorCode ierr = VecDestroy(&(*V)[i]);CHKERRQ(ierr); }
  free(*V); *V = NULL;
  return 0;
}
PetscErrorCode VecDestroy(Vec *v) {
  PetscErrorCode ierr;
  if (!*v) return 0;
  if ((*v)->ops->destroy) { ierr = (*(*v)->ops->destroy)(*v);CHKERRQ(ierr); }
  ierr = PetscLayoutDestroy(&(*v)->map);CHKERRQ(ierr);
  ierr = PetscObjectListDestroy_Private((PetscObject)*v);CHKERRQ(ierr);
  free(*v); *v = NULL;
  return 0;
}
PetscErrorCode VecGetSize(Vec v, PetscInt *N) { VecValid(v, 1); *N = v->map->N; return 0; }
PetscErrorCode VecGetLocalSize(Vec v, PetscInt *n) { VecValid(v, 1); *n = v->map->n; return 0; }
PetscErrorCode VecGetOwnershipRange(Vec v, PetscInt *low, PetscInt *high) {
  VecValid(v, 1);
  if (low) *low = v->map->rstart;
  if (high) *high = v->map->rend;
  return 0;
}
PetscErrorCode VecSetValues(Vec v, PetscInt ni, const PetscInt ix[], const PetscScalar y[], InsertMode mode) {
  VecTypeSet(v, 1);
  PetscErrorCode ierr = (*v->ops->setvalues)(v, ni, ix, y, mode);CHKERRQ(ierr);
  PetscObjectStateIncrease(v);
  return 0;
}
PetscErrorCode VecAssemblyBegin(Vec v) {   /* vector.c:145 */
  VecTypeSet(v, 1);
  if (v->ops->assemblybegin) { PetscErrorCode ierr = (*v->ops->assemblybegin)(v);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode VecAssemblyEnd(Vec v) {
  VecTypeSet(v, 1);
  if (v->ops->assemblyend) { PetscErrorCode ierr = (*v->ops->assemblyend)(v);CHKERRQ(ierr); }
  return 0;
}

/* host access goes through ops->getarray/restorearray, the petscnative == PETSC_FALSE route of
 * include/petsc-private/vecimpl.h:375-385,409-419,430-434 */
PetscErrorCode VecGetArray(Vec v, PetscScalar **a) { VecTypeSet(v, 1); PetscErrorCode ierr = (*v->ops->getarray)(v, a);CHKERRQ(ierr); return 0; }
PetscErrorCode VecRestoreArray(Vec v, PetscScalar **a) {
  VecTypeSet(v, 1);
  PetscErrorCode ierr = (*v->ops->restorearray)(v, a);CHKERRQ(ierr);
  PetscObjectStateIncrease(v);
  return 0;
}
/* the read-only pair takes the same two slots (vecimpl.h:369-400); only VecRestoreArray bumps the state */
PetscErrorCode VecGetArrayRead(Vec v, const PetscScalar **a) { VecTypeSet(v, 1); PetscErrorCode ierr = (*v->ops->getarray)(v, (PetscScalar **)a);CHKERRQ(ierr); return 0; }
PetscErrorCode VecRestoreArrayRead(Vec v, const PetscScalar **a) {
  VecTypeSet(v, 1);
  PetscErrorCode ierr = (*v->ops->restorearray)(v, (PetscScalar **)a);CHKERRQ(ierr);
  if (a) *a = NULL;
  return 0;
}
PetscErrorCode VecPlaceArray(Vec v, const PetscScalar *a) {
  VecTypeSet(v, 1);
  PetscErrorCode ierr = (*v->ops->placearray)(v, a);CHKERRQ(ierr);
  PetscObjectStateIncrease(v);
  return 0;
}
PetscErrorCode VecReplaceArray(Vec v, const PetscScalar *a) {   /* rvector.c:1610-1628 */
  VecTypeSet(v, 1);
  if (!v->ops->replacearray) SETERRQ(v->comm, PETSC_ERR_SUP, "Cannot replace array in this type of vector");
  PetscErrorCode ierr = (*v->ops->replacearray)(v, a);CHKERRQ(ierr);
  PetscObjectStateIncrease(v);
  return 0;
}
PetscErrorCode VecResetArray(Vec v) {
  VecTypeSet(v, 1);
  PetscErrorCode ierr = (*v->ops->resetarray)(v);CHKERRQ(ierr);
  PetscObjectStateIncrease(v);
  return 0;
}

/* ---- rvector.c wrappers ---- */
PetscErrorCode VecSet(Vec x, PetscScalar alpha) {   /* rvector.c:539; caches the norms of a constant vector */
  VecTypeSet(x, 1);
  PetscErrorCode ierr = (*x->ops->set)(x, alpha);CHKERRQ(ierr);
  PetscObjectStateIncrease(x);
  {
    PetscReal val = PetscAbsScalar(alpha);
    x->norm_state[NORM_1] = x->state; x->norm_val[NORM_1] = x->map->N * val;
    x->norm_state[NORM_INFINITY] = x->state; x->norm_val[NORM_INFINITY] = val;
    val = sqrt((double)x->map->N) * val;
    x->norm_state[NORM_2] = x->state; x->norm_val[NORM_2] = val;
  }
  return 0;
}
PetscErrorCode VecCopy(Vec x, Vec y) {   /* vector.c VecCopy: x == y returns */
  VecTypeSet(x, 1); VecTypeSet(y, 2);
  if (x == y) return 0;
  VecSameSize(x, y);
  PetscErrorCode ierr = (*x->ops->copy)(x, y);CHKERRQ(ierr);
  PetscObjectStateIncrease(y);
  for (int t = 0; t < 4; t++) if (x->norm_state[t] == x->state) { y->norm_state[t] = y->state; y->norm_val[t] = x->norm_val[t]; }
  return 0;
}
PetscErrorCode VecSwap(Vec x, Vec y) {
  VecTypeSet(x, 1); VecTypeSet(y, 2); VecSameSize(x, y);
  PetscErrorCode ierr = (*x->ops->swap)(x, y);CHKERRQ(ierr);
  PetscObjectStateIncrease(x); PetscObjectStateIncrease(y);
  return 0;
}
PetscErrorCode VecScale(Vec x, PetscScalar alpha) {   /* rvector.c:476: alpha == 1 returns; known norms are rescaled */
  VecTypeSet(x, 1);
  if (alpha == 1.0) return 0;
  PetscBool have[4]; PetscReal nv[4];
  for (int t = 0; t < 4; t++) { have[t] = (x->norm_state[t] == x->state); nv[t] = x->norm_val[t]; }
  PetscErrorCode ierr = (*x->ops->scale)(x, alpha);CHKERRQ(ierr);
  PetscObjectStateIncrease(x);
  for (int t = 0; t < 4; t++) if (have[t]) { x->norm_state[t] = x->state; x->norm_val[t] = PetscAbsScalar(alpha) * nv[t]; }
  return 0;
}
PetscErrorCode VecAXPY(Vec y, PetscScalar alpha, Vec x) {   /* rvector.c:584 */
  VecTypeSet(x, 3); VecTypeSet(y, 1); VecSameSize(x, y);
  if (x == y) SETERRQ(y->comm, PETSC_ERR_ARG_IDN, "x and y cannot be the same vector");
  PetscErrorCode ierr = (*y->ops->axpy)(y, alpha, x);CHKERRQ(ierr);
  PetscObjectStateIncrease(y);
  return 0;
}
PetscErrorCode VecAYPX(Vec y, PetscScalar alpha, Vec x) {   /* rvector.c:726 */
  VecTypeSet(x, 3); VecTypeSet(y, 1); VecSameSize(x, y);
  if (x == y) SETERRQ(y->comm, PETSC_ERR_ARG_IDN, "x and y must be different vectors");
  PetscErrorCode ierr = (*y->ops->aypx)(y, alpha, x);CHKERRQ(ierr);
  PetscObjectStateIncrease(y);
  return 0;
}
PetscErrorCode VecAXPBY(Vec y, PetscScalar alpha, PetscScalar beta, Vec x) {
  VecTypeSet(x, 4); VecTypeSet(y, 1); VecSameSize(x, y);
  if (x == y) SETERRQ(y->comm, PETSC_ERR_ARG_IDN, "x and y cannot be the same vector");
  PetscErrorCode ierr = (*y->ops->axpby)(y, alpha, beta, x);CHKERRQ(ierr);
  PetscObjectStateIncrease(y);
  return 0;
}
PetscErrorCode VecWAXPY(Vec w, PetscScalar alpha, Vec x, Vec y) {   /* rvector.c:769 */
  VecTypeSet(w, 1); VecTypeSet(x, 3); VecTypeSet(y, 4); VecSameSize(x, y); VecSameSize(w, y);
  if (w == y) SETERRQ(w->comm, PETSC_ERR_SUP, "Result vector w cannot be same as input vector y, suggest VecAXPY()");
  if (w == x) SETERRQ(w->comm, PETSC_ERR_SUP, "Result vector w cannot be same as input vector x, suggest VecAYPX()");
  PetscErrorCode ierr = (*w->ops->waxpy)(w, alpha, x, y);CHKERRQ(ierr);
  PetscObjectStateIncrease(w);
  return 0;
}
PetscErrorCode VecAXPBYPCZ(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y) {
  VecTypeSet(z, 1); VecTypeSet(x, 5); VecTypeSet(y, 6); VecSameSize(x, y); VecSameSize(z, y);
  if (x == y || x == z) SETERRQ(z->comm, PETSC_ERR_ARG_IDN, "x, y, and z must be different vectors");
  if (y == z) SETERRQ(z->comm, PETSC_ERR_ARG_IDN, "x, y, and z must be different vectors");
  PetscErrorCode ierr = (*z->ops->axpbypcz)(z, alpha, beta, gamma, x, y);CHKERRQ(ierr);
  PetscObjectStateIncrease(z);
  return 0;
}
PetscErrorCode VecMAXPY(Vec y, PetscInt nv, const PetscScalar alpha[], Vec x[]) {   /* rvector.c:1220 */
  VecTypeSet(y, 1);
  if (!nv) return 0;
  if (nv < 0) SETERRQ(y->comm, PETSC_ERR_ARG_OUTOFRANGE, "Number of vectors (given %d) cannot be negative", nv);
  for (PetscInt j = 0; j < nv; j++) { VecTypeSet(x[j], 4); VecSameSize(y, x[j]); }
  PetscErrorCode ierr = (*y->ops->maxpy)(y, nv, alpha, x);CHKERRQ(ierr);
  PetscObjectStateIncrease(y);
  return 0;
}
PetscErrorCode VecPointwiseMult(Vec w, Vec x, Vec y) {
  VecTypeSet(w, 1); VecTypeSet(x, 2); VecTypeSet(y, 3); VecSameSize(x, y); VecSameSize(w, y);
  PetscErrorCode ierr = (*w->ops->pointwisemult)(w, x, y);CHKERRQ(ierr);
  PetscObjectStateIncrease(w);
  return 0;
}
PetscErrorCode VecPointwiseDivide(Vec w, Vec x, Vec y) {
  VecTypeSet(w, 1); VecTypeSet(x, 2); VecTypeSet(y, 3); VecSameSize(x, y); VecSameSize(w, y);
  PetscErrorCode ierr = (*w->ops->pointwisedivide)(w, x, y);CHKERRQ(ierr);
  PetscObjectStateIncrease(w);
  return 0;
}
PetscErrorCode VecReciprocal(Vec x) {
  VecTypeSet(x, 1);
  PetscErrorCode ierr = (*x->ops->reciprocal)(x);CHKERRQ(ierr);
  PetscObjectStateIncrease(x);
  return 0;
}
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val) {   /* rvector.c:87 */
  VecTypeSet(x, 1); VecTypeSet(y, 2); VecSameSize(x, y);
  PetscErrorCode ierr = (*x->ops->dot)(x, y, val);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecTDot(Vec x, Vec y, PetscScalar *val) {   /* rvector.c:430 */
  VecTypeSet(x, 1); VecTypeSet(y, 2); VecSameSize(x, y);
  PetscErrorCode ierr = (*x->ops->tdot)(x, y, val);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecMDot(Vec x, PetscInt nv, const Vec y[], PetscScalar val[]) {   /* rvector.c:1173 */
  VecTypeSet(x, 1);
  if (!nv) return 0;
  if (nv < 0) SETERRQ(x->comm, PETSC_ERR_ARG_OUTOFRANGE, "Number of vectors (given %d) cannot be negative", nv);
  for (PetscInt j = 0; j < nv; j++) { VecTypeSet(y[j], 3); VecSameSize(x, y[j]); }
  PetscErrorCode ierr = (*x->ops->mdot)(x, nv, y, val);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecMTDot(Vec x, PetscInt nv, const Vec y[], PetscScalar val[]) { return VecMDot(x, nv, y, val); }

PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val) {   /* rvector.c:193-253 */
  PetscErrorCode ierr;
  VecTypeSet(x, 1);
  if (type == NORM_FROBENIUS) type = NORM_2;
  if (type != NORM_1_AND_2 && x->norm_state[type] == x->state) { *val = x->norm_val[type]; return 0; }
  ierr = (*x->ops->norm)(x, type, val);CHKERRQ(ierr);
  if (type != NORM_1_AND_2) { x->norm_state[type] = x->state; x->norm_val[type] = *val; }
  return 0;
}
PetscErrorCode VecNormalize(Vec x, PetscReal *val) {   /* rvector.c:299 */
  PetscErrorCode ierr;
  PetscReal norm;
  ierr = VecNorm(x, NORM_2, &norm);CHKERRQ(ierr);
  if (norm != 0.0 && norm != 1.0) { ierr = VecScale(x, 1.0 / norm);CHKERRQ(ierr); }
  if (val) *val = norm;
  return 0;
}
PetscErrorCode VecDotNorm2(Vec s, Vec t, PetscScalar *dp, PetscReal *nm) {   /* vinv.c:1200 */
  VecTypeSet(s, 1); VecTypeSet(t, 2); VecSameSize(s, t);
  PetscScalar n2;
  PetscErrorCode ierr = (*s->ops->dotnorm2)(s, t, dp, &n2);CHKERRQ(ierr);
  *nm = n2;
  return 0;
}


/* ---- split-phase reductions (src/vec/vec/utils/comb.c:402-721).  The reference queues the local parts and starts one
 * MPI_Iallreduce in PetscCommSplitReductionBegin; here a vector type may provide the three phases itself
 * ("VecSplitReductionOps_C": its Begin launches the local reduction, the communicator-wide Begin starts the
 * all-reduce, its End delivers); without it Begin simply performs the whole reduction and End hands the value back. */
typedef struct {
  PetscErrorCode (*dot_begin)(Vec, Vec, PetscScalar *);
  PetscErrorCode (*dot_end)(Vec, Vec, PetscScalar *);
  PetscErrorCode (*norm_begin)(Vec, NormType, PetscReal *);
  PetscErrorCode (*norm_end)(Vec, NormType, PetscReal *);
  PetscErrorCode (*comm_begin)(PetscComm);
} VecSplitReductionOps;
static const VecSplitReductionOps *split_ops(Vec x) {
  PetscVoidFunction f = NULL;
  if (PetscObjectQueryFunction((PetscObject)x, "VecSplitReductionOps_C", &f) || !f) return NULL;
  return ((const VecSplitReductionOps *(*)(void))f)();
}
#define SR_MAX 32
static struct { PetscScalar val[SR_MAX]; int n, next; const VecSplitReductionOps *ops; } sr_host = {{0}, 0, 0, NULL};
PetscErrorCode VecDotBegin(Vec x, Vec y, PetscScalar *result) {
  VecTypeSet(x, 1); VecTypeSet(y, 2); VecSameSize(x, y);
  const VecSplitReductionOps *o = split_ops(x);
  if (o) { sr_host.ops = o; return o->dot_begin(x, y, result); }
  if (sr_host.n >= SR_MAX) SETERRQ(x->comm, PETSC_ERR_SUP, "more than %d split reductions in flight", SR_MAX);
  PetscErrorCode ierr = VecDot(x, y, &sr_host.val[sr_host.n++]);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecDotEnd(Vec x, Vec y, PetscScalar *result) {
  const VecSplitReductionOps *o = split_ops(x);
  if (o) return o->dot_end(x, y, result);
  if (sr_host.next >= sr_host.n) SETERRQ(x->comm, PETSC_ERR_ORDER, "VecxxxEnd() without a matching VecxxxBegin()");
  *result = sr_host.val[sr_host.next++];
  if (sr_host.next == sr_host.n) sr_host.next = sr_host.n = 0;
  return 0;
}
PetscErrorCode VecNormBegin(Vec x, NormType type, PetscReal *result) {
  VecTypeSet(x, 1);
  const VecSplitReductionOps *o = split_ops(x);
  if (o) { sr_host.ops = o; return o->norm_begin(x, type, result); }
  if (sr_host.n >= SR_MAX) SETERRQ(x->comm, PETSC_ERR_SUP, "more than %d split reductions in flight", SR_MAX);
  PetscErrorCode ierr = VecNorm(x, type, &sr_host.val[sr_host.n++]);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode VecNormEnd(Vec x, NormType type, PetscReal *result) {
  const VecSplitReductionOps *o = split_ops(x);
  if (o) return o->norm_end(x, type, result);
  if (sr_host.next >= sr_host.n) SETERRQ(x->comm, PETSC_ERR_ORDER, "VecxxxEnd() without a matching VecxxxBegin()");
  *result = sr_host.val[sr_host.next++];
  if (sr_host.next == sr_host.n) sr_host.next = sr_host.n = 0;
  return 0;
}
PetscErrorCode PetscCommSplitReductionBegin(PetscComm comm) {
  if (sr_host.ops && sr_host.ops->comm_begin) return sr_host.ops->comm_begin(comm);
  return 0;
}

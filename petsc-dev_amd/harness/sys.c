/* Harness "Sys": error codes with a traceback string
 * (src/sys/error/err.c), the communicator stand-in, row-block layouts (src/vec/vec/impls/mpi/pmap.c),
 * a string options database (src/sys/objects/options.c) and the flop counter (include/petsclog.h). */
#include "petscimpl.h"
#include <stdarg.h>
#include <ctype.h>

/* ---------------------------------------------------------------- errors */
static char errbuf[4096];
static size_t errlen = 0;

PetscErrorCode PetscError(int line, const char *func, const char *file, PetscErrorCode n, const char *fmt, ...) {
  char msg[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof(msg), fmt, ap);
  va_end(ap);
  if (msg[0] != ' ' || msg[1] != 0) errlen = 0; /* a new error starts a new traceback */
  int w = snprintf(errbuf + errlen, sizeof(errbuf) - errlen, "[%d] %s() line %d in %s %s\n", n, func, line, file, msg);
  if (w > 0 && errlen + (size_t)w < sizeof(errbuf)) errlen += (size_t)w;
  return n ? n : PETSC_ERR_PLIB;
}
const char *PetscGetLastErrorMessage(void) { return errbuf; }

PetscErrorCode PetscMallocFn(size_t bytes, void **p) {
  *p = malloc(bytes ? bytes : 1);
  if (!*p) SETERRQ(0, PETSC_ERR_MEM, "out of memory allocating %zu bytes", bytes);
  return 0;
}

static PetscLogDouble total_flops = 0.0;
PetscErrorCode PetscLogFlops(PetscLogDouble f) { total_flops += f; return 0; }
PetscErrorCode PetscGetFlops(PetscLogDouble *f) { *f = total_flops; return 0; }

/* ---------------------------------------------------------------- communicator */
static struct _p_PetscComm comm_self = {0, 1, NULL, NULL, NULL, NULL, NULL, {NULL, NULL}};
PetscComm PETSC_COMM_SELF = &comm_self;
PetscComm PETSC_COMM_WORLD = &comm_self;

PetscErrorCode PetscCommCreate(int rank, int size, void *ctx, PetscCommAllgatherFn ag, PetscCommAllreduceFn ar,
                               PetscCommBarrierFn bar, PetscComm *comm) {
  PetscErrorCode ierr;
  struct _p_PetscComm *c;
  if (size > 1 && (!ag || !ar)) SETERRQ(0, PETSC_ERR_ARG_NULL, "a communicator of size %d needs allgather and allreduce callbacks", size);
  ierr = PetscMalloc(sizeof(*c), &c);CHKERRQ(ierr);
  c->rank = rank; c->size = size; c->ctx = ctx; c->allgather = ag; c->allreduce = ar; c->barrier = bar; c->exchange = NULL; c->plugin[0] = c->plugin[1] = NULL;
  *comm = c;
  return 0;
}
PetscErrorCode PetscCommSetExchange(PetscComm comm, PetscCommExchangeFn fn) { comm->exchange = fn; return 0; }
PetscErrorCode PetscCommSetWorld(PetscComm comm) { PETSC_COMM_WORLD = comm ? comm : &comm_self; return 0; }
PetscErrorCode PetscCommSetPluginData(PetscComm comm, int slot, void *data) {
  if (slot < 0 || slot > 1) SETERRQ(comm, PETSC_ERR_ARG_OUTOFRANGE, "slot %d", slot);
  comm->plugin[slot] = data;
  return 0;
}
PetscErrorCode PetscCommGetPluginData(PetscComm comm, int slot, void **data) {
  if (slot < 0 || slot > 1) SETERRQ(comm, PETSC_ERR_ARG_OUTOFRANGE, "slot %d", slot);
  *data = comm->plugin[slot];
  return 0;
}
PetscErrorCode PetscCommDestroy(PetscComm *comm) {
  if (*comm && *comm != &comm_self) { if (PETSC_COMM_WORLD == *comm) PETSC_COMM_WORLD = &comm_self; free(*comm); }
  *comm = NULL;
  return 0;
}
PetscErrorCode PetscCommRank(PetscComm comm, PetscMPIInt *rank) { *rank = comm->rank; return 0; }
PetscErrorCode PetscCommSize(PetscComm comm, PetscMPIInt *size) { *size = comm->size; return 0; }

/* ---------------------------------------------------------------- composed functions (src/sys/objects/inherit.c) */
PetscErrorCode PetscObjectComposeFunction(PetscObject obj, const char name[], const char fname[], PetscVoidFunction fn) {
  (void)fname;
  PetscErrorCode ierr;
  struct _n_PetscFList *e;
  if (!obj) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null Object");
  for (e = obj->qlist; e; e = e->next) if (!strcmp(e->name, name)) { e->fn = fn; return 0; }
  ierr = PetscMalloc(sizeof(*e), &e);CHKERRQ(ierr);
  snprintf(e->name, sizeof(e->name), "%s", name);
  e->fn = fn; e->next = obj->qlist; obj->qlist = e;
  return 0;
}
PetscErrorCode PetscObjectQueryFunction(PetscObject obj, const char name[], PetscVoidFunction *fn) {
  *fn = NULL;
  if (!obj) return 0;
  for (struct _n_PetscFList *e = obj->qlist; e; e = e->next) if (!strcmp(e->name, name)) { *fn = e->fn; return 0; }
  return 0;
}
PetscErrorCode PetscObjectListDestroy_Private(PetscObject obj) {
  struct _n_PetscFList *e = obj->qlist;
  while (e) { struct _n_PetscFList *n = e->next; free(e); e = n; }
  obj->qlist = NULL;
  return 0;
}
PetscErrorCode PetscObjectChangeTypeName(PetscObject obj, const char type_name[]) {
  snprintf(obj->type_name, sizeof(obj->type_name), "%s", type_name ? type_name : "");
  return 0;
}

/* ---------------------------------------------------------------- layout */
/* PetscSplitOwnership, src/sys/utils/psplit.c: n = N/size + ((N % size) > rank) */
PetscErrorCode PetscSplitOwnership(PetscComm comm, PetscInt *n, PetscInt *N) {
  if (*N == PETSC_DECIDE && *n == PETSC_DECIDE) SETERRQ(comm, PETSC_ERR_ARG_INCOMP, "Both n and N cannot be PETSC_DECIDE");
  if (*N == PETSC_DECIDE) {
    PetscInt s = *n;
    if (comm->size > 1) { int rc = comm->allreduce(comm->ctx, &s, 1, 0, 0); if (rc) SETERRQ(comm, PETSC_ERR_LIB, "allreduce failed"); }
    *N = s;
  } else if (*n == PETSC_DECIDE) {
    *n = *N / comm->size + ((*N % comm->size) > comm->rank);
  }
  return 0;
}

/* PetscLayoutSetUp, src/vec/vec/impls/mpi/pmap.c: gather every rank's n, prefix-sum into range[] */
PetscErrorCode PetscLayoutCreateSetUp(PetscComm comm, PetscInt n, PetscInt N, PetscLayout *map) {
  PetscErrorCode ierr;
  PetscLayout m;
  ierr = PetscSplitOwnership(comm, &n, &N);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(*m), &m);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(comm->size + 1), &m->range);CHKERRQ(ierr);
  m->n = n; m->N = N; m->refcnt = 1;
  if (comm->size > 1) {
    PetscInt *all;
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)comm->size, &all);CHKERRQ(ierr);
    if (comm->allgather(comm->ctx, &n, (int)sizeof(PetscInt), all)) SETERRQ(comm, PETSC_ERR_LIB, "allgather failed");
    m->range[0] = 0;
    for (int p = 0; p < comm->size; p++) m->range[p + 1] = m->range[p] + all[p];
    free(all);
    if (m->range[comm->size] != N) SETERRQ(comm, PETSC_ERR_ARG_SIZ, "Sum of local lengths %d does not equal global length %d", m->range[comm->size], N);
  } else { m->range[0] = 0; m->range[1] = n; if (N != n) SETERRQ(comm, PETSC_ERR_ARG_SIZ, "local size %d != global size %d on one process", n, N); }
  m->rstart = m->range[comm->rank]; m->rend = m->range[comm->rank + 1];
  *map = m;
  return 0;
}
PetscErrorCode PetscLayoutReference(PetscLayout in, PetscLayout *out) { in->refcnt++; *out = in; return 0; }
PetscErrorCode PetscLayoutDestroy(PetscLayout *map) {
  if (*map && --(*map)->refcnt == 0) { free((*map)->range); free(*map); }
  *map = NULL;
  return 0;
}

/* ---------------------------------------------------------------- options */
#define MAXOPT 128
static char opt_name[MAXOPT][64], opt_val[MAXOPT][128];
static int nopt = 0;

PetscErrorCode PetscOptionsClear(void) { nopt = 0; return 0; }
PetscErrorCode PetscOptionsSetValue(const char *name, const char *value) {
  if (name[0] == '-') name++;
  for (int i = 0; i < nopt; i++) if (!strcmp(opt_name[i], name)) { snprintf(opt_val[i], sizeof(opt_val[i]), "%s", value ? value : ""); return 0; }
  if (nopt >= MAXOPT) SETERRQ(0, PETSC_ERR_PLIB, "options table full");
  snprintf(opt_name[nopt], sizeof(opt_name[nopt]), "%s", name);
  snprintf(opt_val[nopt], sizeof(opt_val[nopt]), "%s", value ? value : "");
  nopt++;
  return 0;
}
/* "-a 1 -b -c foo": an option's value is the next token unless that token starts with '-' followed by a letter */
PetscErrorCode PetscOptionsInsertString(const char *str) {
  PetscErrorCode ierr;
  char buf[2048], *tok[256];
  int nt = 0;
  snprintf(buf, sizeof(buf), "%s", str);
  for (char *p = strtok(buf, " \t\n"); p && nt < 256; p = strtok(NULL, " \t\n")) tok[nt++] = p;
  for (int i = 0; i < nt; i++) {
    if (tok[i][0] != '-') continue;
    const char *name = tok[i], *val = "";
    if (i + 1 < nt && !(tok[i + 1][0] == '-' && isalpha((unsigned char)tok[i + 1][1]))) val = tok[++i];
    ierr = PetscOptionsSetValue(name, val);CHKERRQ(ierr);
  }
  return 0;
}
PetscErrorCode PetscOptionsGetString(const char *pre, const char *name, char *value, size_t len, PetscBool *set) {
  char full[128];
  if (name[0] == '-') name++;
  snprintf(full, sizeof(full), "%s%s", pre ? pre : "", name);
  *set = PETSC_FALSE;
  for (int i = 0; i < nopt; i++) if (!strcmp(opt_name[i], full)) { snprintf(value, len, "%s", opt_val[i]); *set = PETSC_TRUE; }
  return 0;
}
PetscErrorCode PetscOptionsGetInt(const char *pre, const char *name, PetscInt *value, PetscBool *set) {
  char v[128];
  PetscErrorCode ierr = PetscOptionsGetString(pre, name, v, sizeof(v), set);CHKERRQ(ierr);
  if (*set) *value = (PetscInt)atol(v);
  return 0;
}
PetscErrorCode PetscOptionsGetReal(const char *pre, const char *name, PetscReal *value, PetscBool *set) {
  char v[128];
  PetscErrorCode ierr = PetscOptionsGetString(pre, name, v, sizeof(v), set);CHKERRQ(ierr);
  if (*set) *value = atof(v);
  return 0;
}

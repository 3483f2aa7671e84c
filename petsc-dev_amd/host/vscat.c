/* HipScatter, parallel -> sequential "general" form: the halo exchange of MatMult_MPIAIJ.
 * Set-up restates VecScatterCreate_PtoS (src/vec/vec/utils/vpscat.c:1730-1924) with the index-list
 * exchange done by one all-gather of every rank's request list instead of Isend/Irecv pairs (same
 * resulting to/from lists, bit for bit).  Begin/End replace HipScatterBegin_1/End_1
 * (vpscat.h:14-233: pack, persistent MPI_Start, MPI_Waitany, unpack) by
 *     device pack kernel -> RCCL grouped send/recv over xGMI on the halo stream -> device unpack,
 * ordered against the compute stream with two HIP events so the diagonal-block SpMV overlaps it. */
#include "hipmi355ximpl.h"

static int owner_of(int size, const PetscInt *range, PetscInt idx) {   /* vpscat.c:1762-1772 */
  for (int j = 0; j < size; j++) if (idx < range[j + 1]) return j;
  return -1;
}

static PetscBool is_contiguous(const PetscInt *idx, PetscInt n) {
  for (PetscInt k = 1; k < n; k++) if (idx[k] != idx[0] + k) return PETSC_FALSE;
  return PETSC_TRUE;
}

PetscErrorCode HipScatterCreate_PtoS_MPIAIJ(MPI_Comm comm, PetscLayout xmap, PetscInt ec, const PetscInt *garray, HipScatter *out) {
  PetscErrorCode ierr;
  HipScatter ctx;
  int size = HipCommSize(comm), rank = HipCommRank(comm);
  PetscInt *ecs, maxec = 0, *all = NULL;
  ierr = PetscMalloc(sizeof(*ctx), &ctx);CHKERRQ(ierr);
  memset(ctx, 0, sizeof(*ctx));
  ctx->comm = comm;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)size, &ecs);CHKERRQ(ierr);
  if (size > 1) {
    if (HipCommAllgather(comm, &ec, (int)sizeof(PetscInt), ecs)) SETERRQ(comm, PETSC_ERR_LIB, "allgather failed");
  } else ecs[0] = ec;
  for (int p = 0; p < size; p++) maxec = PetscMax(maxec, ecs[p]);
  if (size > 1 && maxec > 0) {
    PetscInt *mine;
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)maxec, &mine);CHKERRQ(ierr);
    memset(mine, 0, sizeof(PetscInt) * (size_t)maxec);
    memcpy(mine, garray, sizeof(PetscInt) * (size_t)ec);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)maxec * (size_t)size, &all);CHKERRQ(ierr);
    if (HipCommAllgather(comm, mine, (int)(sizeof(PetscInt) * (size_t)maxec), all)) SETERRQ(comm, PETSC_ERR_LIB, "allgather failed");
    HipFree(mine);
  }
  /* ---- "from" (receive) side: owners ascending, slots in order of appearance (vpscat.c:1871-1885) ---- */
  VecScatterSide *from = &ctx->from, *to = &ctx->to;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(size + 1), &from->procs);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(size + 2), &from->starts);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(ec, 1), &from->indices);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(ec, 1), &from->local_slots);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(ec, 1), &to->local_slots);CHKERRQ(ierr);
  PetscInt cnt = 0;
  from->n = 0; from->starts[0] = 0;
  for (int p = 0; p < size; p++) {
    if (p == rank) continue;
    PetscBool have = PETSC_FALSE;
    for (PetscInt i = 0; i < ec; i++) {
      int o = owner_of(size, xmap->range, garray[i]);
      if (o < 0) SETERRQ(comm, PETSC_ERR_PLIB, "ith %d block entry %d not owned by any process, upper bound %d", i, garray[i], xmap->range[size]);
      if (o == p) { from->indices[cnt++] = i; have = PETSC_TRUE; }
    }
    if (have) { from->procs[from->n++] = p; from->starts[from->n] = cnt; }
  }
  /* local part (vpscat.c:1897-1910) */
  PetscInt nl = 0;
  for (PetscInt i = 0; i < ec; i++)
    if (garray[i] >= xmap->range[rank] && garray[i] < xmap->range[rank + 1]) { to->local_slots[nl] = garray[i] - xmap->range[rank]; from->local_slots[nl++] = i; }
  from->local_n = to->local_n = nl;
  /* ---- "to" (send) side: requesting ranks ascending, their order of request (vpscat.c:1782,1846-1856) ---- */
  PetscInt total = 0;
  for (int q = 0; q < size; q++) if (q != rank) total += ecs[q];
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(size + 1), &to->procs);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(size + 2), &to->starts);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(total, 1), &to->indices);CHKERRQ(ierr);
  cnt = 0; to->n = 0; to->starts[0] = 0;
  for (int q = 0; q < size; q++) {
    if (q == rank) continue;
    PetscBool have = PETSC_FALSE;
    const PetscInt *gq = all + (size_t)q * (size_t)maxec;
    for (PetscInt i = 0; i < ecs[q]; i++)
      if (gq[i] >= xmap->range[rank] && gq[i] < xmap->range[rank + 1]) { to->indices[cnt++] = gq[i] - xmap->range[rank]; have = PETSC_TRUE; }
    if (have) { to->procs[to->n++] = q; to->starts[to->n] = cnt; }
  }
  HipFree(all); HipFree(ecs);
  /* contiguity (the to->contiq / from->contiq special case, vpscat.c:1951-1960), here per side */
  from->contiq = PETSC_TRUE;
  for (PetscInt i = 0; i < from->n; i++) if (!is_contiguous(from->indices + from->starts[i], from->starts[i + 1] - from->starts[i])) from->contiq = PETSC_FALSE;
  to->contiq = PETSC_TRUE;
  for (PetscInt i = 0; i < to->n; i++) if (!is_contiguous(to->indices + to->starts[i], to->starts[i + 1] - to->starts[i])) to->contiq = PETSC_FALSE;
  *out = ctx;
  return 0;
}

PetscErrorCode HipScatterGetLists(HipScatter ctx, PetscInt *nrecv, const PetscInt **rprocs, const PetscInt **rstarts, const PetscInt **rindices,
                                  PetscInt *nsend, const PetscInt **sprocs, const PetscInt **sstarts, const PetscInt **sindices,
                                  PetscInt *nlocal, const PetscInt **lto, const PetscInt **lfrom) {
  if (!ctx) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null HipScatter");
  *nrecv = ctx->from.n; *rprocs = ctx->from.procs; *rstarts = ctx->from.starts; *rindices = ctx->from.indices;
  *nsend = ctx->to.n; *sprocs = ctx->to.procs; *sstarts = ctx->to.starts; *sindices = ctx->to.indices;
  *nlocal = ctx->to.local_n; *lto = ctx->to.local_slots; *lfrom = ctx->from.local_slots;
  return 0;
}

/* one-time index upload (VecScatterInitializeForGPU, src/vec/vec/utils/veccusp/vscatcusp.c:29-112) */
static PetscErrorCode scatter_device_setup(HipScatter ctx, PetscDeviceCtx *dc) {
  if (ctx->device_ready) return 0;
  VecScatterSide *s[2] = {&ctx->to, &ctx->from};
  for (int k = 0; k < 2; k++) {
    PetscInt tot = s[k]->starts[s[k]->n];
    if (tot > 0) {
      CHKHIP(mi355x_malloc((void **)&s[k]->d_indices, sizeof(PetscInt) * (size_t)tot));
      CHKHIP(mi355x_memcpy_h2d(dc->h, s[k]->d_indices, s[k]->indices, sizeof(PetscInt) * (size_t)tot));
      CHKHIP(mi355x_malloc((void **)&s[k]->d_values, sizeof(PetscScalar) * (size_t)tot));
    }
    if (s[k]->local_n > 0) {
      CHKHIP(mi355x_malloc((void **)&s[k]->d_local_slots, sizeof(PetscInt) * (size_t)s[k]->local_n));
      CHKHIP(mi355x_memcpy_h2d(dc->h, s[k]->d_local_slots, s[k]->local_slots, sizeof(PetscInt) * (size_t)s[k]->local_n));
    }
  }
  if (ctx->to.local_n > 0) CHKHIP(mi355x_malloc((void **)&ctx->d_local_tmp, sizeof(PetscScalar) * (size_t)ctx->to.local_n));
  CHKHIP(mi355x_handle_synchronize(dc->h));
  CHKHIP(mi355x_event_create(&ctx->ev_packed));
  CHKHIP(mi355x_event_create(&ctx->ev_done));
  ctx->device_ready = 1;
  return 0;
}

/* One neighbour exchange: receives into rdst[i], sends from ssrc[i] (device pointers).  RCCL: one grouped
 * ncclRecv/ncclSend on the halo stream.  Without an RCCL communicator (several ranks sharing one GPU in the
 * rehearsal tests) the same buffers travel device -> host -> launcher-supplied exchange -> host -> device, the
 * arrangement of the reference's own CUSP path (vpscat.h:56-61,226-230); everything around the transport --
 * pack, contiguity shortcuts, offsets, unpack order -- is shared. */
static PetscErrorCode neighbour_exchange(HipScatter ctx, PetscDeviceCtx *dc, PetscInt nr, const PetscInt *rprocs, PetscScalar *const *rdst, const PetscInt *rcnt,
                                         PetscInt ns, const PetscInt *sprocs, const PetscScalar *const *ssrc, const PetscInt *scnt) {
  PetscErrorCode ierr;
  MPI_Comm comm = ctx->comm;
  if (HipCommDevice(comm)) {
    int rc = 0, rc_end;
    CHKHIP(mi355x_comm_group_start());
    for (PetscInt i = 0; i < nr && !rc; i++) rc = mi355x_comm_recv(HipCommDeviceHalo(comm), dc->hcomm, rdst[i], (size_t)rcnt[i], rprocs[i]);
    for (PetscInt i = 0; i < ns && !rc; i++) rc = mi355x_comm_send(HipCommDeviceHalo(comm), dc->hcomm, ssrc[i], (size_t)scnt[i], sprocs[i]);
    rc_end = mi355x_comm_group_end();                       /* always closed, also after a failed post */
    CHKHIP(rc);
    CHKHIP(rc_end);
    return 0;
  }
  if (!HipCommHasExchange(comm)) SETERRQ(comm, PETSC_ERR_ORDER, "parallel scatter needs PetscCommSetDeviceComm() (RCCL) or PetscCommSetExchange() (host-staged)");
  size_t stot = 0, rtot = 0;
  for (PetscInt i = 0; i < ns; i++) stot += (size_t)scnt[i];
  for (PetscInt i = 0; i < nr; i++) rtot += (size_t)rcnt[i];
  if (!ctx->h_send) { ierr = PetscMalloc(sizeof(PetscScalar) * PetscMax((size_t)ctx->to.starts[ctx->to.n] + (size_t)ctx->from.starts[ctx->from.n], 1), &ctx->h_send);CHKERRQ(ierr); }
  if (!ctx->h_recv) { ierr = PetscMalloc(sizeof(PetscScalar) * PetscMax((size_t)ctx->to.starts[ctx->to.n] + (size_t)ctx->from.starts[ctx->from.n], 1), &ctx->h_recv);CHKERRQ(ierr); }
  /* per-neighbour arrays out of the scatter's work block (scatter_nbr_work: 10 arrays of max(to.n, from.n) entries; 4..9 are ours) */
  const size_t nmax = (size_t)PetscMax(PetscMax(ctx->to.n, ctx->from.n), 1);
  void **sb = (void **)ctx->nbr_work + 4 * nmax, **rb = sb + nmax;
  int *sp = (int *)(rb + nmax), *rp = sp + nmax, *sbytes = rp + nmax, *rbytes = sbytes + nmax;
  size_t off = 0;
  for (PetscInt i = 0; i < ns; i++) {
    CHKHIP(mi355x_memcpy_d2h(dc->hcomm, ctx->h_send + off, ssrc[i], sizeof(PetscScalar) * (size_t)scnt[i]));
    sb[i] = ctx->h_send + off; sp[i] = sprocs[i]; sbytes[i] = (int)(sizeof(PetscScalar) * (size_t)scnt[i]);
    off += (size_t)scnt[i];
  }
  off = 0;
  for (PetscInt i = 0; i < nr; i++) { rb[i] = ctx->h_recv + off; rp[i] = rprocs[i]; rbytes[i] = (int)(sizeof(PetscScalar) * (size_t)rcnt[i]); off += (size_t)rcnt[i]; }
  CHKHIP(mi355x_handle_synchronize(dc->hcomm));
  if (HipCommExchange(comm, (int)ns, sp, sb, sbytes, (int)nr, rp, rb, rbytes)) SETERRQ(comm, PETSC_ERR_LIB, "host exchange failed");
  for (PetscInt i = 0; i < nr; i++) CHKHIP(mi355x_memcpy_h2d(dc->hcomm, rdst[i], rb[i], (size_t)rbytes[i]));
  CHKHIP(mi355x_handle_synchronize(dc->hcomm));   /* the staging buffer is pageable and reused */
  return 0;
}

/* Records "the source vector is final" on the compute stream NOW, so that the caller may queue independent work
 * (the diagonal-block SpMV) on the compute stream before HipScatterBegin spends host time enqueueing the RCCL
 * operations: the halo stream still only waits for what preceded this point. */
PetscErrorCode HipScatterMarkReady(HipScatter ctx, Vec x) {
  PetscErrorCode ierr;
  PetscDeviceCtx *dc;
  const PetscScalar *dx;
  if (!ctx || (ctx->to.n == 0 && ctx->from.n == 0 && ctx->to.local_n == 0)) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);   /* a host-side x is uploaded before the mark, not after it */
  ierr = scatter_device_setup(ctx, dc);CHKERRQ(ierr);
  CHKHIP(mi355x_event_record(ctx->ev_packed, dc->h));
  ctx->ready_marked = 1;
  return 0;
}

/* FORWARD: x (parallel) -> y (= lvec, sequential); REVERSE: x (= lvec) -> y (parallel).  INSERT_VALUES, ADD_VALUES and MAX_VALUES in
 * either direction (UnPack_1's three forms, vpscat.c:503-534); MatMult_MPIAIJ uses FORWARD/INSERT, the transpose REVERSE/ADD. */
static int unpack_mode(mi355x_handle_t h, InsertMode addv, size_t n, const PetscInt *idx, const PetscScalar *buf, PetscScalar *y) {
  return addv == ADD_VALUES ? mi355x_unpack_add(h, n, idx, buf, y) : addv == MAX_VALUES ? mi355x_unpack_max(h, n, idx, buf, y) : mi355x_unpack_insert(h, n, idx, buf, y);
}
static PetscErrorCode scatter_begin(HipScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode);
PetscErrorCode HipScatterBegin(HipScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode) {
  PetscErrorCode ierr;
  if (!ctx) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null HipScatter");
  if (ctx->inuse) SETERRQ(ctx->comm, PETSC_ERR_ARG_WRONGSTATE, " Scatter ctx already in use");   /* vscat.c:1637 */
  if (addv != INSERT_VALUES && addv != ADD_VALUES && addv != MAX_VALUES) SETERRQ(ctx->comm, PETSC_ERR_ARG_WRONG, "Cannot handle insert mode %d", (int)addv);   /* vpscat.c:530 */
  if (mode != SCATTER_FORWARD && mode != SCATTER_REVERSE) SETERRQ(ctx->comm, PETSC_ERR_SUP, "SCATTER_LOCAL forms are not ported");
  ierr = scatter_begin(ctx, x, y, addv, mode);CHKERRQ(ierr);
  ctx->inuse = PETSC_TRUE;                                  /* only a Begin that succeeded leaves the scatter in use */
  return 0;
}
static PetscErrorCode scatter_begin(HipScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode) {
  PetscErrorCode ierr;
  PetscDeviceCtx *dc;
  if (ctx->to.n == 0 && ctx->from.n == 0 && ctx->to.local_n == 0) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = scatter_device_setup(ctx, dc);CHKERRQ(ierr);
  /* any number of neighbours: the per-neighbour arrays live in a work block of the scatter (8-byte slots; arrays 0..3 here, 4..9
   * in neighbour_exchange) */
  const size_t nmax = (size_t)PetscMax(PetscMax(ctx->to.n, ctx->from.n), 1);
  if (!ctx->nbr_work) { ierr = PetscMalloc(sizeof(void *) * 10 * nmax, &ctx->nbr_work);CHKERRQ(ierr); }
  PetscScalar **rdst = (PetscScalar **)ctx->nbr_work; const PetscScalar **ssrc = (const PetscScalar **)ctx->nbr_work + nmax;
  PetscInt *rcnt = (PetscInt *)((void **)ctx->nbr_work + 2 * nmax), *scnt = (PetscInt *)((void **)ctx->nbr_work + 3 * nmax);
  VecScatterSide *to = &ctx->to, *from = &ctx->from;
  if (mode == SCATTER_FORWARD) {
    const PetscScalar *dx; PetscScalar *dy;
    ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
    ierr = VecHIPGetReadWrite(y, &dy);CHKERRQ(ierr);
    /* the halo stream starts after everything queued on the compute stream when x became final */
    if (!ctx->ready_marked) CHKHIP(mi355x_event_record(ctx->ev_packed, dc->h));
    ctx->ready_marked = 0;
    CHKHIP(mi355x_handle_wait_event(dc->hcomm, ctx->ev_packed));
    const PetscBool timed = (PetscBool)(ctx->timing && ctx->time_n < ctx->time_cap);
    if (timed) CHKHIP(mi355x_event_record(ctx->time_ev[2 * ctx->time_n], dc->hcomm));
    PetscInt nsend = to->starts[to->n];
    if (nsend && !to->contiq) CHKHIP(mi355x_pack(dc->hcomm, (size_t)nsend, to->d_indices, dx, to->d_values));   /* Pack_1 */
    if (to->n || from->n) {
      for (PetscInt i = 0; i < from->n; i++) {
        PetscInt s = from->starts[i];
        rcnt[i] = from->starts[i + 1] - s;
        rdst[i] = (from->contiq && addv == INSERT_VALUES) ? dy + from->indices[s] : from->d_values + s;   /* in place only when the values replace */
      }
      for (PetscInt i = 0; i < to->n; i++) {
        PetscInt s = to->starts[i];
        scnt[i] = to->starts[i + 1] - s;
        ssrc[i] = to->contiq ? dx + to->indices[s] : to->d_values + s;
      }
      ierr = neighbour_exchange(ctx, dc, from->n, from->procs, rdst, rcnt, to->n, to->procs, ssrc, scnt);CHKERRQ(ierr);
    }
    if (from->n && !(from->contiq && addv == INSERT_VALUES)) CHKHIP(unpack_mode(dc->hcomm, addv, (size_t)from->starts[from->n], from->d_indices, from->d_values, dy));   /* UnPack_1 */
    if (to->local_n) {   /* Scatter_1, vpscat.c:538; staged through the scatter's own buffer */
      CHKHIP(mi355x_pack(dc->hcomm, (size_t)to->local_n, to->d_local_slots, dx, ctx->d_local_tmp));
      CHKHIP(unpack_mode(dc->hcomm, addv, (size_t)to->local_n, from->d_local_slots, ctx->d_local_tmp, dy));
    }
    if (timed) { CHKHIP(mi355x_event_record(ctx->time_ev[2 * ctx->time_n + 1], dc->hcomm)); ctx->time_n++; }
    CHKHIP(mi355x_event_record(ctx->ev_done, dc->hcomm));
  } else {
    const PetscScalar *dx; PetscScalar *dy;
    ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
    ierr = VecHIPGetReadWrite(y, &dy);CHKERRQ(ierr);
    CHKHIP(mi355x_event_record(ctx->ev_packed, dc->h));
    CHKHIP(mi355x_handle_wait_event(dc->hcomm, ctx->ev_packed));
    PetscInt nback = from->starts[from->n];
    if (nback && !from->contiq) CHKHIP(mi355x_pack(dc->hcomm, (size_t)nback, from->d_indices, dx, from->d_values));
    if (to->n || from->n) {
      for (PetscInt i = 0; i < to->n; i++) {
        PetscInt s = to->starts[i];
        rcnt[i] = to->starts[i + 1] - s;
        rdst[i] = to->d_values + s;
      }
      for (PetscInt i = 0; i < from->n; i++) {
        PetscInt s = from->starts[i];
        scnt[i] = from->starts[i + 1] - s;
        ssrc[i] = from->contiq ? dx + from->indices[s] : from->d_values + s;
      }
      ierr = neighbour_exchange(ctx, dc, to->n, to->procs, rdst, rcnt, from->n, from->procs, ssrc, scnt);CHKERRQ(ierr);
    }
    CHKHIP(mi355x_event_record(ctx->ev_done, dc->hcomm));
  }
  return 0;
}

PetscErrorCode HipScatterEnd(HipScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode) {
  PetscErrorCode ierr;
  PetscDeviceCtx *dc;
  if (!ctx) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null HipScatter");
  ctx->inuse = PETSC_FALSE;
  if (ctx->to.n == 0 && ctx->from.n == 0 && ctx->to.local_n == 0) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  /* the compute stream resumes after the halo stream's work (replaces MPI_Waitany, vpscat.h:210) */
  CHKHIP(mi355x_handle_wait_event(dc->h, ctx->ev_done));
  if (mode == SCATTER_REVERSE) {
    /* y[idx] (+)= received, one neighbour after the other in rank order: the updates happen here,
     * after the local transpose product, as mpiaij.c:1162-1164 assumes; the order is fixed (the
     * reference's is arrival order unless -vecscatter_reproduce, vpscat.h:206-208) */
    PetscScalar *dy;
    VecScatterSide *to = &ctx->to, *from = &ctx->from;
    ierr = VecHIPGetReadWrite(y, &dy);CHKERRQ(ierr);
    for (PetscInt i = 0; i < to->n; i++) {
      PetscInt s = to->starts[i], c = to->starts[i + 1] - s;
      CHKHIP(unpack_mode(dc->h, addv, (size_t)c, to->d_indices + s, to->d_values + s, dy));
    }
    if (to->local_n) {
      const PetscScalar *dx;
      ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
      CHKHIP(mi355x_pack(dc->h, (size_t)to->local_n, from->d_local_slots, dx, ctx->d_local_tmp));
      CHKHIP(unpack_mode(dc->h, addv, (size_t)to->local_n, to->d_local_slots, ctx->d_local_tmp, dy));
    }
  }
  ierr = VecHIPRestoreWrite(y);CHKERRQ(ierr);
  HipStateIncrease(y);
  return 0;
}

/* bench.py: an event pair on the halo stream around every forward exchange from now on (on), read by the MPIAIJ matrix that owns the
 * scatter (MatMPIAIJHIPMI355XGetHaloTiming) */
PetscErrorCode HipScatterSetTiming(HipScatter ctx, PetscBool on) {
  if (!ctx) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null HipScatter");
  ctx->timing = on; ctx->time_n = 0;
  if (on && !ctx->time_ev) {
    ctx->time_cap = 4096;
    PetscErrorCode ierr = PetscMalloc(sizeof(mi355x_event_t) * 2 * (size_t)ctx->time_cap, &ctx->time_ev);CHKERRQ(ierr);
    memset(ctx->time_ev, 0, sizeof(mi355x_event_t) * 2 * (size_t)ctx->time_cap);
    for (PetscInt k = 0; k < 2 * ctx->time_cap; k++) CHKHIP(mi355x_event_create(&ctx->time_ev[k]));
  }
  return 0;
}

PetscErrorCode HipScatterDestroy(HipScatter *pctx) {
  HipScatter ctx = *pctx;
  if (!ctx) return 0;
  if (ctx->time_ev) { for (PetscInt k = 0; k < 2 * ctx->time_cap; k++) if (ctx->time_ev[k]) mi355x_event_destroy(ctx->time_ev[k]); HipFree(ctx->time_ev); }
  VecScatterSide *s[2] = {&ctx->to, &ctx->from};
  for (int k = 0; k < 2; k++) {
    HipFree(s[k]->procs); HipFree(s[k]->starts); HipFree(s[k]->indices); HipFree(s[k]->local_slots);
    if (s[k]->d_indices) mi355x_free(s[k]->d_indices);
    if (s[k]->d_values) mi355x_free(s[k]->d_values);
    if (s[k]->d_local_slots) mi355x_free(s[k]->d_local_slots);
  }
  if (ctx->d_local_tmp) mi355x_free(ctx->d_local_tmp);
  if (ctx->ev_packed) mi355x_event_destroy(ctx->ev_packed);
  if (ctx->ev_done) mi355x_event_destroy(ctx->ev_done);
  HipFree(ctx->h_send); HipFree(ctx->h_recv); HipFree(ctx->nbr_work);
  HipFree(ctx);
  *pctx = NULL;
  return 0;
}

#if !defined(PETSCHIPMI355X_WITH_PETSC)
/* PETSc's names for the same calls: on the harness this scatter is the only VecScatter there is */
PetscErrorCode VecScatterBegin(VecScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode) { return HipScatterBegin(ctx, x, y, addv, mode); }
PetscErrorCode VecScatterEnd(VecScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode) { return HipScatterEnd(ctx, x, y, addv, mode); }
PetscErrorCode VecScatterDestroy(VecScatter *ctx) { return HipScatterDestroy(ctx); }
#endif
PetscErrorCode VecScatterGetLists(VecScatter ctx, PetscInt *nrecv, const PetscInt **rprocs, const PetscInt **rstarts, const PetscInt **rindices,
                                  PetscInt *nsend, const PetscInt **sprocs, const PetscInt **sstarts, const PetscInt **sindices,
                                  PetscInt *nlocal, const PetscInt **lto, const PetscInt **lfrom) {
  return HipScatterGetLists((HipScatter)ctx, nrecv, rprocs, rstarts, rindices, nsend, sprocs, sstarts, sindices, nlocal, lto, lfrom);
}

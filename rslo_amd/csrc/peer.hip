// Same-stream sum over the ranks of one node for the SyncBN statistics (include/rslo_hip.h: rslo_peer_*).
//
// The reference normalises the BEV head with apex SyncBatchNorm (rslo/layers/SparseConv.py:96-132, bn_type "SyncBN",
// train_hdf5.py:463): 45 layers, i.e. 90 exchanges of 2C(+1) doubles per training step, each BETWEEN two dependent kernels
// of the same layer.  As RCCL collectives they cost a collective launch plus two stream hand-offs inside
// ProcessGroupNCCL each, 90 times on the critical path.  Here an exchange is ONE small kernel on the training stream:
//
//   every rank owns a SLICE of memory every other rank of the node can read:  [SLOTS] x { payload granules | channel records }
//   exchange number q (the same on every rank: they run the same layers in the same order) uses slot q mod SLOTS:
//     1. write the own payload into the own slice as 8-byte granules { half of a double | tag = q } (system-scope stores)
//     2. every thread polls the granules of ITS element in every slice until they show tag q (system-scope loads, s_sleep,
//        bounded by a timeout) and adds them IN RANK ORDER -> the same bits on every rank; no flag, no fence
//   Slot reuse: a rank can write exchange q + 1 only after it has finished reading q, and nobody can finish q + 1 without
//   its granules, so no rank is ever more than one exchange ahead of a reader: 2 slots suffice, 4 are used.
//
// Two transports behind the same kernel (a table of slice base pointers):
//   * host:   one POSIX shared-memory segment holding all slices, mapped and hipHostRegister-ed by every rank (fine-grained,
//             system-coherent host memory; every access is a PCIe transaction).  Works for any set of GPUs of one host --
//             also for several ranks on ONE GPU, which is how the single-GPU development box tests it.
//   * device: each rank's slice in its own HBM (hipExtMallocWithFlags fine-grained), exported with hipIpcGetMemHandle and
//             opened by the peers: polls and pulls are xGMI reads of <= 4 KB per peer.
// Nothing here is a collective library: one node, <= 16 ranks, <= 1024 doubles per exchange.
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <new>

#include "peer_comm.h"


// Device-memory transport, fence-free since round 5: element i travels as two 8-byte granules { 32-bit half of the double | 32-bit tag = exchange
// number }, each written and polled with ONE system-scope 8-byte access, so a granule that shows the tag IS the data (the
// release / acquire fences of the flag protocol wrote back / invalidated the XCD's L2 once per exchange, right behind a
// convolution that left it dirty).
__global__ void __launch_bounds__(1024) k_peer_allreduce(double *__restrict__ t, int n, PeerTable tab, int me, int world,
                                                         PeerSeq sq, long long timeout_ticks,
                                                         unsigned long long *__restrict__ status, unsigned *__restrict__ wait_ring) {
  const int tid = threadIdx.x;
  const unsigned long long seq = peer_seq_value(sq);
  const size_t slot_off = (size_t)(seq % PEER_SLOTS) * sq.slot_bytes;
  const unsigned tag = (unsigned)seq;
  __shared__ int s_bad;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(t[i]);
    unsigned long long *g = (unsigned long long *)(tab.base[me] + slot_off + 64) + 2 * i;
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(g + 1, ((unsigned long long)tag << 32) | (unsigned)(bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  unsigned waited = 0;
  for (int i = tid; i < n; i += blockDim.x) {
    double s = 0.0;
    for (int r = 0; r < world; ++r) {                      // rank order: the same bits on every rank
      const unsigned long long *g = (const unsigned long long *)(tab.base[r] + slot_off + 64) + 2 * i;
      unsigned long long w0, w1;
      const long long t0 = wall_clock64();
      while (true) {
        w0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        w1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag) break;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > timeout_ticks) {
          s_bad = r + 1;
          break;
        }
      }
      if (r != me) waited = max(waited, (unsigned)(wall_clock64() - t0));
      s += __longlong_as_double((long long)((w1 << 32) | (unsigned)w0));
    }
    t[i] = s;
  }
  if (tid == 0) __hip_atomic_store(wait_ring + (seq % PEER_WAIT_RING), waited, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (s_bad) {       // a peer never arrived: poison the result and report; never hang the GPU
    if (tid == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
      __hip_atomic_store(status + 1, (unsigned long long)(s_bad - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(status, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int i = tid; i < n; i += blockDim.x) t[i] = __builtin_nan("");
  }
}

// Host-memory transport: every access is a PCIe transaction, so the payload goes as plain doubles behind ONE flag (bulk
// stores, release fence, flag; the peers spin on the flag only): 9.3 us per exchange of 513 doubles against 129 us for
// the granule form there.  The system-scope fences write back / invalidate the XCD's L2, which is why the device transport
// does not use this form.
__global__ void __launch_bounds__(1024) k_peer_allreduce_flag(double *__restrict__ t, int n, PeerTable tab, int me, int world,
                                                              PeerSeq sq, long long timeout_ticks,
                                                              unsigned long long *__restrict__ status, unsigned *__restrict__ wait_ring) {
  const int tid = threadIdx.x;
  const unsigned long long seq = peer_seq_value(sq);
  const size_t slot_off = (size_t)(seq % PEER_SLOTS) * sq.slot_bytes;
  unsigned char *mine = tab.base[me] + slot_off;
  unsigned long long *my_flag = (unsigned long long *)mine;
  double *my_pay = (double *)(mine + 64);
  for (int i = tid; i < n; i += blockDim.x)
    __hip_atomic_store(my_pay + i, t[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (tid == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);      // system scope: the payload is visible before the flag
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(my_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __shared__ int s_bad;
  __shared__ unsigned s_wait;
  if (tid == 0) { s_bad = 0; s_wait = 0; }
  __syncthreads();
  if (tid < world && tid != me) {
    const unsigned long long *f = (const unsigned long long *)(tab.base[tid] + slot_off);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > timeout_ticks) {
        s_bad = tid + 1;
        break;
      }
    }
    atomicMax(&s_wait, (unsigned)(wall_clock64() - t0));
  }
  __syncthreads();
  if (tid == 0) __hip_atomic_store(wait_ring + (seq % PEER_WAIT_RING), s_wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope
  if (s_bad) {       // a peer never arrived: poison the result and report; never hang the GPU
    if (tid == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
      __hip_atomic_store(status + 1, (unsigned long long)(s_bad - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(status, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int i = tid; i < n; i += blockDim.x) t[i] = __builtin_nan("");
    return;
  }
  for (int i = tid; i < n; i += blockDim.x) {
    double s = 0.0;
    for (int r = 0; r < world; ++r) {
      const double *p = (const double *)(tab.base[r] + slot_off + 64);
      s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    t[i] = s;
  }
}

static int peer_common_init(RsloPeerComm *c, int rank, int world, int max_n) {
  c->rank = rank; c->world = world; c->max_n = max_n;
  c->slot_bytes = peer_slot_bytes(max_n);
  c->slice_bytes = c->slot_bytes * PEER_SLOTS;
  c->seq = 0;
  c->timeout_ticks = (long long)600 * 100000000LL;      // 600 s at 100 MHz: the scale of a process-group timeout (rslo_peer_set_timeout_ms)
  RSLO_HIP(hipHostMalloc((void **)&c->status_host, 64, hipHostMallocMapped));
  memset(c->status_host, 0, 64);
  RSLO_HIP(hipHostGetDevicePointer((void **)&c->status_dev, c->status_host, 0));
  RSLO_HIP(hipHostMalloc((void **)&c->wait_ring_host, PEER_WAIT_RING * sizeof(unsigned), hipHostMallocMapped));
  memset(c->wait_ring_host, 0, PEER_WAIT_RING * sizeof(unsigned));
  RSLO_HIP(hipHostGetDevicePointer((void **)&c->wait_ring_dev, c->wait_ring_host, 0));
  RSLO_HIP(hipMalloc((void **)&c->seq_word_dev, 64));
  RSLO_HIP(hipMemset(c->seq_word_dev, 0, 64));
  c->capturing = 0;
  c->cap_count = 0;
  return RSLO_OK;
}

extern "C" int rslo_peer_create_host(const char *name, int rank, int world, int max_n, void **comm_out) {
  RSLO_CHECK_ARG(name && comm_out && world >= 1 && world <= PEER_MAX_WORLD && rank >= 0 && rank < world && max_n >= 1 &&
                     max_n <= PEER_MAX_N && strlen(name) < 120,
                 "rslo_peer_create_host: bad arguments (world <= %d, max_n <= %d)", PEER_MAX_WORLD, PEER_MAX_N);
  RsloPeerComm *c = new (std::nothrow) RsloPeerComm();
  RSLO_CHECK_ARG(c, "rslo_peer_create_host: out of memory");
  memset(c, 0, sizeof(*c));
  c->transport = 0;
  int rc = peer_common_init(c, rank, world, max_n);
  if (rc != RSLO_OK) { delete c; return rc; }
  snprintf(c->shm_name, sizeof(c->shm_name), "%s", name);
  c->shm_bytes = ((c->slice_bytes * world + 4095) / 4096) * 4096;
  // every rank opens-or-creates the same segment; a new segment is zero-filled by the kernel (flags 0 = nothing sent: sequence
  // numbers start at 1); sizing it twice to the same size is harmless
  int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->shm_bytes) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    rslo_set_error("rslo_peer_create_host: shm_open/ftruncate(%s) failed", name);
    return RSLO_EINVAL;
  }
  c->shm_ptr = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->shm_ptr == MAP_FAILED) {
    delete c;
    rslo_set_error("rslo_peer_create_host: mmap failed");
    return RSLO_EINVAL;
  }
  hipError_t e = hipHostRegister(c->shm_ptr, c->shm_bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  void *dptr = nullptr;
  if (e == hipSuccess) e = hipHostGetDevicePointer(&dptr, c->shm_ptr, 0);
  if (e != hipSuccess) {
    munmap(c->shm_ptr, c->shm_bytes);
    delete c;
    rslo_set_error("rslo_peer_create_host: hipHostRegister: %s", hipGetErrorString(e));
    return RSLO_ELAUNCH;
  }
  for (int r = 0; r < world; ++r) c->tab.base[r] = (unsigned char *)dptr + (size_t)r * c->slice_bytes;
  *comm_out = c;
  return RSLO_OK;
}

// Once every rank has the segment mapped (the caller's barrier), its NAME can go: the mapping stays alive while mapped, and a
// rank that dies later leaves nothing behind in /dev/shm (which is memory on a box without swap).
extern "C" int rslo_peer_host_unlink(void *comm) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && c->transport == 0, "rslo_peer_host_unlink: not a host-transport comm");
  if (c->shm_name[0]) {
    shm_unlink(c->shm_name);
    c->shm_name[0] = 0;
  }
  return RSLO_OK;
}

extern "C" int rslo_peer_ipc_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int rslo_peer_create_device_begin(int rank, int world, int max_n, void **comm_out, void *handle_out) {
  RSLO_CHECK_ARG(comm_out && handle_out && world >= 1 && world <= PEER_MAX_WORLD && rank >= 0 && rank < world &&
                     max_n >= 1 && max_n <= PEER_MAX_N,
                 "rslo_peer_create_device_begin: bad arguments");
  RsloPeerComm *c = new (std::nothrow) RsloPeerComm();
  RSLO_CHECK_ARG(c, "rslo_peer_create_device_begin: out of memory");
  memset(c, 0, sizeof(*c));
  c->transport = 1;
  int rc = peer_common_init(c, rank, world, max_n);
  if (rc != RSLO_OK) { delete c; return rc; }
  hipError_t e = hipExtMallocWithFlags(&c->own_slice, c->slice_bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(c->own_slice, 0, c->slice_bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle((hipIpcMemHandle_t *)handle_out, c->own_slice);
  if (e != hipSuccess) {
    if (c->own_slice) (void)hipFree(c->own_slice);
    delete c;
    rslo_set_error("rslo_peer_create_device_begin: %s", hipGetErrorString(e));
    return RSLO_ELAUNCH;
  }
  c->tab.base[rank] = (unsigned char *)c->own_slice;
  *comm_out = c;
  return RSLO_OK;
}

extern "C" int rslo_peer_create_device_finish(void *comm, const void *all_handles) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && all_handles && c->transport == 1, "rslo_peer_create_device_finish: bad arguments");
  const hipIpcMemHandle_t *h = (const hipIpcMemHandle_t *)all_handles;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h[r], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      rslo_set_error("rslo_peer_create_device_finish: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
      return RSLO_ELAUNCH;
    }
    c->peer_open[r] = p;
    c->tab.base[r] = (unsigned char *)p;
  }
  return RSLO_OK;
}

extern "C" int rslo_peer_set_timeout_ms(void *comm, int ms) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && ms >= 1, "rslo_peer_set_timeout_ms: bad arguments");
  c->timeout_ticks = (long long)ms * 100000LL;
  return RSLO_OK;
}

extern "C" int rslo_peer_allreduce_f64(void *comm, double *t, int n, void *stream) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && t && n >= 1 && n <= c->max_n, "rslo_peer_allreduce_f64: n = %d outside 1..%d", n, c ? c->max_n : 0);
  const PeerSeq sq = peer_next_seq(c);      // committed only once the launch is in the stream
  const int threads = n >= 512 ? 1024 : (n >= 192 ? 512 : 256);
  if (!c->capturing) c->wait_ring_host[sq.seq % PEER_WAIT_RING] = 0;
  if (c->transport == 0)
    hipLaunchKernelGGL(k_peer_allreduce_flag, dim3(1), dim3(threads), 0, (hipStream_t)stream, t, n, c->tab, c->rank, c->world,
                       sq, c->timeout_ticks, c->status_dev, c->wait_ring_dev);
  else
    hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(threads), 0, (hipStream_t)stream, t, n, c->tab, c->rank, c->world, sq,
                       c->timeout_ticks, c->status_dev, c->wait_ring_dev);
  RSLO_CHECK_LAUNCH("k_peer_allreduce");
  peer_commit_seq(c);
  return RSLO_OK;
}

// ---- exchanges inside a replayed stream capture (rslo_amd/headgraph.py: the BEV head's forward as one hipGraph) ---------------
// Every rank captures the same layers in the same order, so exchange k of a replay is number base + k on every rank, base = the
// number of exchanges the rank had issued when the replay was launched.  The kernels of the capture read base from a device
// word; rslo_peer_replay_prepare writes it in stream order (a one-thread kernel whose argument is the host's counter) and
// advances the host's counter past the replay's exchanges -- launches issued eagerly before and after a replay (the backward
// pass) keep their absolute numbers, and the order of all exchanges on the stream is the order of their numbers, as ever.
__global__ void k_peer_set_word(unsigned long long *word, unsigned long long value) { *word = value; }

extern "C" int rslo_peer_capture_begin(void *comm) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && !c->capturing, "rslo_peer_capture_begin: no comm, or a capture is already open");
  c->capturing = 1;
  c->cap_count = 0;
  return RSLO_OK;
}

extern "C" int rslo_peer_capture_end(void *comm, int *n_exchanges) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && c->capturing, "rslo_peer_capture_end: no capture is open");
  c->capturing = 0;
  if (n_exchanges) *n_exchanges = (int)c->cap_count;
  return RSLO_OK;
}

extern "C" int rslo_peer_replay_prepare(void *comm, int n_exchanges, void *stream) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && !c->capturing && n_exchanges >= 0 && n_exchanges <= PEER_WAIT_RING,
                 "rslo_peer_replay_prepare: bad arguments (or a capture is open)");
  if (n_exchanges == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_peer_set_word, dim3(1), dim3(1), 0, (hipStream_t)stream, c->seq_word_dev, c->seq);
  RSLO_CHECK_LAUNCH("k_peer_set_word");
  for (int k = 1; k <= n_exchanges; ++k) c->wait_ring_host[(c->seq + k) % PEER_WAIT_RING] = 0;
  c->seq += (unsigned long long)n_exchanges;
  return RSLO_OK;
}

// 0 = every exchange so far met all its peers; else the sequence number of the first one that timed out (*peer = the rank
// that never arrived).  Reads pinned memory: no synchronisation, the answer lags the stream.
extern "C" unsigned long long rslo_peer_status(void *comm, int *peer) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  if (!c) return ~0ULL;
  const unsigned long long s = ((volatile unsigned long long *)c->status_host)[0];
  if (peer) *peer = s ? (int)((volatile unsigned long long *)c->status_host)[1] : -1;
  return s;
}

// Diagnostics: the waits (microseconds) of the most recent exchanges, oldest first -- how long each spent spinning for its
// slowest peer (arrival skew + transport latency; 0 for a one-rank comm).  Reads pinned memory: synchronise the stream first for
// a consistent picture.  Returns the number of samples written (<= max_out, <= PEER_WAIT_RING, <= exchanges issued).
extern "C" int rslo_peer_wait_samples(void *comm, float *out_us, int max_out) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  if (!c || !out_us || max_out <= 0) return 0;
  unsigned long long n = c->seq < PEER_WAIT_RING ? c->seq : PEER_WAIT_RING;
  if (n > (unsigned long long)max_out) n = (unsigned long long)max_out;
  for (unsigned long long k = 0; k < n; ++k) {
    const unsigned long long q = c->seq - n + 1 + k;
    out_us[k] = (float)((volatile unsigned *)c->wait_ring_host)[q % PEER_WAIT_RING] * 0.01f;      // 100 MHz ticks
  }
  return (int)n;
}

extern "C" int rslo_peer_destroy(void *comm) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  if (!c) return RSLO_OK;
  (void)hipDeviceSynchronize();
  if (c->transport == 0) {
    (void)hipHostUnregister(c->shm_ptr);
    munmap(c->shm_ptr, c->shm_bytes);
    if (c->rank == 0 && c->shm_name[0]) shm_unlink(c->shm_name);
  } else {
    for (int r = 0; r < c->world; ++r)
      if (c->peer_open[r]) (void)hipIpcCloseMemHandle(c->peer_open[r]);
    if (c->own_slice) (void)hipFree(c->own_slice);
  }
  (void)hipHostFree(c->status_host);
  (void)hipHostFree(c->wait_ring_host);
  if (c->seq_word_dev) (void)hipFree(c->seq_word_dev);
  delete c;
  return RSLO_OK;
}

// The peer communicator shared by peer.hip (stand-alone exchange kernel) and bn2d.hip (SyncBN kernels that exchange their
// per-channel sums themselves): layout of a rank's slice and the per-channel rendezvous every block of such a kernel runs.
#pragma once
#include "rslo_common.h"

#define PEER_SLOTS 4
#define PEER_MAX_WORLD 16
#define PEER_MAX_N 1024
#define PEER_WAIT_RING 4096              // per-exchange wait samples kept for the diagnostics (rslo_peer_wait_samples)
#define PEER_MAX_CH 512                 // channels of a fused SyncBN exchange (per-channel flags)
#define PEER_CH_BYTES 64                // six 8-byte granules {32-bit half | 32-bit tag} per channel (3 doubles), padded

struct PeerTable {
  unsigned char *base[PEER_MAX_WORLD];      // slice of every rank, as THIS process addresses it
};

struct RsloPeerComm {
  int rank, world, max_n, transport;        // 0 host shm, 1 device ipc
  size_t slice_bytes, slot_bytes;
  unsigned long long seq;                   // exchanges issued so far
  // Exchanges launched inside a stream capture (rslo_peer_capture_begin .. _end) carry their number RELATIVE to a device word
  // the host sets in stream order in front of every replay (rslo_peer_replay_prepare): number = *seq_word + k, k = 1, 2, ... in
  // capture order.  Their launch arguments are then replay-invariant.
  unsigned long long *seq_word_dev;
  int capturing;
  unsigned long long cap_count;             // exchanges launched since rslo_peer_capture_begin
  PeerTable tab;
  unsigned long long *status_host;          // pinned: [0] = first sequence number that timed out (0 = none), [1] = peer
  unsigned long long *status_dev;
  long long timeout_ticks;                  // wall_clock64() ticks (100 MHz)
  unsigned *wait_ring_host, *wait_ring_dev; // pinned [PEER_WAIT_RING]: ticks exchange q spent waiting for its peers, at q mod ring
  // host transport
  void *shm_ptr;
  size_t shm_bytes;
  char shm_name[128];
  // device transport
  void *own_slice;
  void *peer_open[PEER_MAX_WORLD];
};


// a slot = { 64 bytes unused | payload [max_n] x 2 granules { half u32 | tag u32 } | per-channel records [PEER_MAX_CH] x 6 granules { half u32 | tag u32 } }
static inline size_t peer_chan_off(int max_n) { return 64 + (size_t)max_n * 2 * sizeof(unsigned long long); }
static inline size_t peer_slot_bytes(int max_n) { return peer_chan_off(max_n) + (size_t)PEER_MAX_CH * PEER_CH_BYTES; }

// The exchange number as a kernel sees it: absolute (word == NULL) or relative to the device word of a replayed capture.
struct PeerSeq {
  unsigned long long seq;
  const unsigned long long *word;
  size_t slot_bytes;
};
__device__ __forceinline__ unsigned long long peer_seq_value(const PeerSeq &q) { return q.seq + (q.word ? *q.word : 0ULL); }
// (host) the number the next launch carries, and its commit once the launch is in the stream
static inline PeerSeq peer_next_seq(const RsloPeerComm *c) {
  PeerSeq q;
  q.seq = c->capturing ? c->cap_count + 1 : c->seq + 1;
  q.word = c->capturing ? c->seq_word_dev : nullptr;
  q.slot_bytes = c->slot_bytes;
  return q;
}
static inline void peer_commit_seq(RsloPeerComm *c) {
  if (c->capturing) ++c->cap_count;
  else ++c->seq;
}

// Per-channel rendezvous of ONE workgroup with the workgroups of the same channel on the other ranks (fused SyncBN,
// bn2d.hip).  v[3] in: this rank's values of channel c (valid in thread 0); out: their sums over the ranks IN RANK ORDER,
// broadcast to every thread through `sh`.  chan = byte offset of the slot's channel records.  Returns false when a peer
// did not arrive within the timeout (status written; the caller poisons its outputs).  The workgroups of different
// channels never wait for each other, only for their peers on the other GPUs, which need nothing from this GPU to get
// there: placement-independent, no co-residency assumption.
// NO FENCES: a channel's record is six 8-byte granules { 32-bit half of a double | 32-bit tag = exchange number }, each
// written and polled with ONE system-scope 8-byte access (atomic by itself), so a granule that shows the tag IS the data.
// A release / acquire pair at system scope would write back / invalidate the XCD's L2 -- per WORKGROUP here, right after a
// convolution left it full of dirty lines: measured 15.2 ms per step against 13.3 for the three-launch path and 11.7 for
// local BatchNorm (profiles/NOTES.md round 5).
__device__ __forceinline__ bool peer_chan_exchange(const PeerTable &tab, int me, int world, unsigned long long seq, size_t chan,
                                                   int c, long long timeout_ticks, unsigned long long *status, double (&v)[3],
                                                   double *sh /* [4 + 4 * PEER_MAX_WORLD] shared */, unsigned *wait_ring = nullptr) {
  const int tid = threadIdx.x;
  const unsigned tag = (unsigned)seq;
  unsigned *halves = (unsigned *)(sh + 4);                 // [world][6]
  if (tid == 0) {
    sh[0] = v[0]; sh[1] = v[1]; sh[2] = v[2];
    sh[3] = 0.0;
  }
  __syncthreads();
  if (tid < 6) {                                          // publish: one granule per thread
    const unsigned long long bits = (unsigned long long)__double_as_longlong(sh[tid >> 1]);
    const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
    unsigned long long *g = (unsigned long long *)(tab.base[me] + chan + (size_t)c * PEER_CH_BYTES) + tid;
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid < 6 * world) {                                  // collect: thread (rank r, granule k) polls until the tag shows
    const int r = tid / 6, k = tid - 6 * r;
    const unsigned long long *g = (const unsigned long long *)(tab.base[r] + chan + (size_t)c * PEER_CH_BYTES) + k;
    const long long t0 = wall_clock64();
    unsigned long long w;
    while ((unsigned)((w = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 32) != tag) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > timeout_ticks) {
        sh[3] = (double)(r + 1);
        break;
      }
    }
    halves[r * 6 + k] = (unsigned)w;
    if (k == 0) halves[6 * world + r] = r != me ? (unsigned)(wall_clock64() - t0) : 0u;
  }
  __syncthreads();
  // diagnostics: how long channel 0's workgroup waited for the slowest OTHER rank (one sample per exchange, plain store)
  if (wait_ring && c == 0 && tid == 0) {
    unsigned mx = 0;
    for (int r = 0; r < world; ++r) mx = halves[6 * world + r] > mx ? halves[6 * world + r] : mx;
    __hip_atomic_store(wait_ring + (seq % PEER_WAIT_RING), mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int bad = (int)sh[3];
  if (bad) {
    if (tid == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
      __hip_atomic_store(status + 1, (unsigned long long)(bad - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(status, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return false;
  }
  if (tid < 3) {
    double s = 0.0;
    for (int r = 0; r < world; ++r)
      s += __longlong_as_double((long long)(((unsigned long long)halves[r * 6 + 2 * tid + 1] << 32) | halves[r * 6 + 2 * tid]));
    sh[tid] = s;
  }
  __syncthreads();
  v[0] = sh[0]; v[1] = sh[1]; v[2] = sh[2];
  return true;
}

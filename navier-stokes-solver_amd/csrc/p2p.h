// Peer-to-peer transport of the row-partitioned loops over xGMI without a collective library on the critical path:
// every rank owns a small FINE-GRAINED region (mailbox + halo flags + landing zone) that its peers map through HIP IPC
// and write into with plain remote stores.
//
//   all-reduce of one double   inside the sum kernel that produced the local sum: lanes r < nranks store the value,
//                              split into two 8-byte words (32 bits of payload + the 32-bit sequence number each: an
//                              8-byte store is single-copy atomic, no fence needed), into slot [seq & 1][my rank] of peer
//                              r's mailbox, then spin on their own mailbox until the word pair of rank r carries this
//                              sequence number; thread 0 adds the nranks values IN RANK ORDER -- the same bits on every
//                              rank.  No extra launch, no host involvement; one xGMI store latency + the skew of the ranks.
//   halo exchange              "put": a small kernel copies the boundary runs of the operand straight into the landing
//                              zones of the neighbours (remote stores), fences, and the last workgroup of a segment
//                              raises the neighbour's flag to the sequence number; "wait + copy": every workgroup spins
//                              on the flags of this rank's sources, then moves its share of the landing zone behind the
//                              owned entries of the operand.
// Every spin is bounded (wall clock): a peer that never arrives stops the loop with an error code instead of hanging
// the GPU.  Two mailbox parities suffice: a rank cannot finish all-reduce q + 1 before every rank has entered it, i.e.
// left all-reduce q.  Every landing zone exists TWICE, used alternately by the exchanges of its layout: a neighbour may
// run one exchange ahead (BPCG v1 and MINRES exchange twice in a row without an all-reduce in between: its put of
// exchange e + 1 can arrive before this rank has copied exchange e out -- found by running the native BPCG v1 loop with
// two ranks), never two: its put of exchange e + 2 follows its wait for e + 1, i.e. this rank's put of e + 1, which this
// rank's stream orders behind its copy of e.
#pragma once

#include "nss_common.h"

#include <vector>

namespace nss {

constexpr int kP2pMaxRanks = 16;
constexpr int kP2pMaxSegments = 8;
constexpr int kP2pMaxChannels = 4;    // operand layouts one handle serves (BPCG v2: t1; MINRES / BPCG v1: A's and B^T's operand)
constexpr unsigned long long kP2pTimeoutTicks = 300000000ull;   // wall_clock64 runs at 100 MHz: 3 s

struct P2pView {                      // what the kernels need (passed by value)
  unsigned long long* mail = nullptr;             // own mailbox: [2][nranks][2] words
  unsigned long long* const* peer_mail = nullptr; // DEVICE array [nranks]: the peers' mailboxes (own entry: `mail`)
  int32_t nranks = 1, rank = 0;
  uint32_t seq = 0;                   // sequence number of THIS collective (host counter, same order on every rank)
  int32_t* error = nullptr;           // device word: set to 1 on a timeout
};

__device__ __forceinline__ unsigned long long p2p_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void p2p_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Sum of `local` (taken from thread 0) over the ranks, evaluated by the calling workgroup (>= nranks threads, ALL must
// call); the result is valid in every thread.  A timeout sets the error word (the caller stops the loop).
// `lds`: >= kP2pMaxRanks doubles.
__device__ __forceinline__ double p2p_allreduce_sum(const P2pView& v, double local, double* lds) {
  const int t = threadIdx.x;
  __syncthreads();
  if (t == 0) lds[0] = local;
  __syncthreads();
  const double mine = lds[0];
  __syncthreads();
  const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
  const unsigned long long tag = (unsigned long long)v.seq << 32;
  const int par = int(v.seq & 1u);
  if (t < v.nranks) {
    unsigned long long* dst = v.peer_mail[t] + (size_t(par) * v.nranks + v.rank) * 2;
    p2p_store(dst, (bits & 0xffffffffull) | tag);
    p2p_store(dst + 1, (bits >> 32) | tag);
    const unsigned long long* src = v.mail + (size_t(par) * v.nranks + t) * 2;
    const unsigned long long t0 = wall_clock64();
    unsigned long long lo = p2p_load(src), hi = p2p_load(src + 1);
    while ((lo >> 32) != v.seq || (hi >> 32) != v.seq) {
      if (wall_clock64() - t0 > kP2pTimeoutTicks) {
        atomicExch(v.error, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      lo = p2p_load(src);
      hi = p2p_load(src + 1);
    }
    lds[t] = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
  }
  __syncthreads();
  double s = 0.0;
  for (int r = 0; r < v.nranks; ++r) s += lds[r];        // rank order: identical bits on every rank
  return s;
}

}  // namespace nss

// host side of the transport (p2p.hip)
struct nss_p2p_s {
  int32_t nranks = 1, rank = 0;
  char* region = nullptr;                    // own fine-grained region
  size_t region_bytes = 0;
  unsigned long long* mail = nullptr;        // region + 0
  std::vector<char*> peer_region;            // mapped peers (own: region)
  unsigned long long** d_peer_mail = nullptr;   // device array [nranks]
  int32_t* d_error = nullptr;
  int32_t* d_ticket = nullptr;               // [kP2pMaxSegments] workgroup tickets of the put kernel
  uint32_t seq = 0;                          // host counter of collectives issued
  // One CHANNEL per operand layout this handle serves (bound at creation): its arrival flags (one per source rank),
  // its landing zone, the own receive table and, after connect, where the peers want our segments.  A halo descriptor
  // finds its channel by the identity of its host tables (h_send_off / h_recv_off): the loops pass copies of ONE
  // descriptor per layout with only `ext` changed.
  struct Channel {
    const void *key_send = nullptr, *key_recv = nullptr;
    int32_t n_owned = 0;
    size_t flags_off = 0, landing_off = 0;   // byte offsets inside the own region (flags_off is the same in every region)
    int64_t landing_doubles = 0;
    size_t zone_bytes = 0;                   // distance of the two copies of the own landing zone
    mutable uint32_t count = 0;              // exchanges of this layout issued so far (host; the same on every rank)
    std::vector<int64_t> peer_zone_bytes;    // per destination rank: the distance of ITS two zones
    std::vector<int64_t> recv_off, recv_cnt; // per source rank: offset into the landing zone / count (0: none)
    std::vector<int64_t> peer_land_off;      // per destination rank: BYTE offset inside ITS region where our segment goes
  };
  std::vector<Channel> channels;
  bool connected = false;
  const Channel* find(const nss_halo_t& h) const {
    for (const Channel& c : channels)
      if (c.key_send == (const void*)h.h_send_off && c.key_recv == (const void*)h.h_recv_off) return &c;
    return nullptr;
  }
  nss::P2pView view(uint32_t s) const { return nss::P2pView{mail, d_peer_mail, nranks, rank, s, d_error}; }
};

namespace nss {
// halo exchange of `h` (direct sends only; one of the layouts the handle was created for) through its landing zone:
// put + wait/copy on stream `st`
void p2p_exchange(nss_p2p_s& p, const nss_halo_t& h, const int32_t* done, hipStream_t st);
// dst[0] = sum over the ranks of src[0], the ranks' values added in rank order (one small launch)
void p2p_allreduce(nss_p2p_s& p, const double* src, double* dst, hipStream_t st);
}  // namespace nss

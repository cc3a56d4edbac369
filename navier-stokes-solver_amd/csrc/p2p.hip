// Peer-to-peer transport over xGMI through IPC-mapped mailboxes and landing zones (see p2p.h).
#include "p2p.h"
#include "dist.h"

#include <cstring>

namespace nss {

struct PutSeg {
  const double* src;            // first entry of the run inside this rank's operand
  double* dst;                  // where it goes: inside the neighbour's landing zone (remote)
  unsigned long long* flag;     // the neighbour's arrival flag of this rank (remote)
  int32_t cnt, wg0, nwg;        // doubles; first workgroup / workgroups of this segment in the launch
};
struct PutArgs {
  PutSeg seg[kP2pMaxSegments];
  int32_t nseg;
  uint32_t seq;
  int32_t* ticket;
  const int32_t* done;
};

constexpr int kPutPerWg = kBlock * 8;

__global__ __launch_bounds__(kBlock) void p2p_put_kernel(PutArgs a) {
  if (a.done && a.done[0] != 0) return;
  int sidx = 0;
  for (int i = 1; i < a.nseg; ++i)
    if (int(blockIdx.x) >= a.seg[i].wg0) sidx = i;
  const PutSeg& g = a.seg[sidx];
  const int base = (int(blockIdx.x) - g.wg0) * kPutPerWg;
  for (int i = base + int(threadIdx.x); i < base + kPutPerWg && i < g.cnt; i += kBlock)
    __builtin_nontemporal_store(g.src[i], g.dst + i);            // remote store over xGMI
  __threadfence_system();                                         // the data before the flag
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = atomicAdd(a.ticket + sidx, 1);
    if (t == g.nwg - 1) {                                          // last workgroup of this segment
      a.ticket[sidx] = 0;
      p2p_store(g.flag, (unsigned long long)a.seq);
    }
  }
}

struct WaitArgs {
  const unsigned long long* flags;   // own flags, one per source rank
  int32_t src_rank[kP2pMaxSegments];
  int32_t nsrc;
  uint32_t seq;
  const double* landing;
  double* dst;                       // operand + n_owned
  int64_t n;
  int32_t* error;
  const int32_t* done;
};

__global__ __launch_bounds__(kBlock) void p2p_wait_copy_kernel(WaitArgs a) {
  if (a.done && a.done[0] != 0) return;
  if (int(threadIdx.x) < a.nsrc) {
    const unsigned long long* f = a.flags + a.src_rank[threadIdx.x];
    const unsigned long long t0 = wall_clock64();
    while (p2p_load(f) < (unsigned long long)a.seq) {
      if (wall_clock64() - t0 > kP2pTimeoutTicks) {
        atomicExch(a.error, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __threadfence_system();                                         // the flag before the data it announces
  __syncthreads();
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < a.n; i += stride)
    a.dst[i] = __builtin_nontemporal_load(a.landing + i);         // fine-grained memory: not cached
}

void p2p_exchange(nss_p2p_s& p, const nss_halo_t& h, const int32_t* done, hipStream_t st) {
  if (!p.connected) throw Error("p2p: not connected");
  if (h.n_send == 0 && h.n_recv == 0) return;                     // no neighbour (one rank)
  if (h.n_send > 0 && !h.direct) throw Error("p2p: the halo must send contiguous runs of the operand (direct)");
  if (h.n_send > kP2pMaxSegments || h.n_recv > kP2pMaxSegments) throw Error("p2p: too many neighbours");
  const nss_p2p_s::Channel* ch = p.find(h);
  if (!ch) throw Error("p2p: this halo layout was not registered when the transport was created");
  const uint32_t seq = ++p.seq;
  const size_t par = size_t(ch->count++ & 1u);                  // which of the two landing zones this exchange uses
  if (h.n_send > 0) {
    PutArgs a{};
    int wg = 0;
    for (int i = 0; i < h.n_send; ++i) {
      const int peer = h.h_send_peer[i];
      const int64_t cnt = h.h_send_cnt[i];
      if (ch->peer_land_off[size_t(peer)] < 0) throw Error("p2p: a neighbour does not expect our segment");
      char* pr = p.peer_region[size_t(peer)];
      unsigned long long* pflags = reinterpret_cast<unsigned long long*>(pr + ch->flags_off);   // (same offset in every region)
      double* pdst = reinterpret_cast<double*>(pr + ch->peer_land_off[size_t(peer)] +          // byte offset the peer published
                                               par * size_t(ch->peer_zone_bytes[size_t(peer)]));
      a.seg[i] = PutSeg{h.ext + h.h_send_off[i], pdst, pflags + p.rank, int32_t(cnt), wg,
                        int((cnt + kPutPerWg - 1) / kPutPerWg)};
      wg += a.seg[i].nwg;
    }
    a.nseg = h.n_send;
    a.seq = seq;
    a.ticket = p.d_ticket;
    a.done = done;
    hipLaunchKernelGGL(p2p_put_kernel, dim3(wg), dim3(kBlock), 0, st, a);
    NSS_CHECK_LAUNCH();
  }
  if (h.n_recv > 0) {
    WaitArgs w{};
    int64_t total = 0;
    for (int i = 0; i < h.n_recv; ++i) {
      w.src_rank[i] = h.h_recv_peer[i];
      if (h.h_recv_off[i] - ch->n_owned != total) throw Error("p2p: receive segments must fill the ghost tail in order");
      total += h.h_recv_cnt[i];
    }
    if (total > ch->landing_doubles) throw Error("p2p: landing zone too small");
    w.flags = reinterpret_cast<const unsigned long long*>(p.region + ch->flags_off);
    w.nsrc = h.n_recv;
    w.seq = seq;
    w.landing = reinterpret_cast<const double*>(p.region + ch->landing_off + par * ch->zone_bytes);
    w.dst = h.ext + ch->n_owned;
    w.n = total;
    w.error = p.d_error;
    w.done = done;
    hipLaunchKernelGGL(p2p_wait_copy_kernel, dim3(stream_grid(total, kBlock * 4)), dim3(kBlock), 0, st, w);
    NSS_CHECK_LAUNCH();
  }
}

__global__ __launch_bounds__(kBlock) void p2p_allreduce_kernel(P2pView v, const double* src, double* dst) {
  __shared__ double lds[kP2pMaxRanks];
  const double s = p2p_allreduce_sum(v, threadIdx.x == 0 ? src[0] : 0.0, lds);
  if (threadIdx.x == 0) dst[0] = s;
}

void p2p_allreduce(nss_p2p_s& p, const double* src, double* dst, hipStream_t st) {
  if (!p.connected) throw Error("p2p: not connected");
  hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(kBlock), 0, st, p.view(++p.seq), src, dst);
  NSS_CHECK_LAUNCH();
}

}  // namespace nss

using namespace nss;

namespace {
size_t mail_bytes(int nranks) { return sizeof(unsigned long long) * 2 * size_t(nranks) * 2; }
size_t flags_bytes() { return sizeof(unsigned long long) * size_t(kP2pMaxRanks); }
size_t chan_blob(int nranks) { return size_t(16) * nranks + 8; }   // where[nranks] | count[nranks] | zone_bytes
size_t blob_bytes(int nranks, int nchan) { return 64 + chan_blob(nranks) * nchan; }
}  // namespace

extern "C" {

// blob layout (what every rank publishes): 64 bytes IPC handle | per channel: int64 byte offset (inside the publisher's
// region) where rank q's segment goes, q < nranks (-1: nothing); int64 count from rank q; int64 distance of its two zones
int nss_p2p_blob_bytes(int32_t nranks, int32_t nhalo, int64_t* bytes) {
  return guarded([&] {
    NSS_REQUIRE(nranks >= 1 && nranks <= kP2pMaxRanks && nhalo >= 1 && nhalo <= kP2pMaxChannels && bytes, "p2p_blob_bytes: bad argument");
    *bytes = int64_t(blob_bytes(nranks, nhalo));
  });
}

int nss_p2p_create(int32_t nranks, int32_t rank, int32_t nhalo, const nss_halo_t* const* halos, const int32_t* n_owned,
                   nss_p2p_t* out, void* h_blob) {
  return guarded([&] {
    NSS_REQUIRE(out && h_blob && halos && n_owned, "p2p_create: NULL argument");
    NSS_REQUIRE(nranks >= 1 && nranks <= kP2pMaxRanks && rank >= 0 && rank < nranks, "p2p_create: bad rank / size (at most 16 ranks)");
    NSS_REQUIRE(nhalo >= 1 && nhalo <= kP2pMaxChannels, "p2p_create: 1 .. 4 operand layouts");
    nss_p2p_s* p = new nss_p2p_s;
    try {
      p->nranks = nranks;
      p->rank = rank;
      size_t off = mail_bytes(nranks);
      for (int c = 0; c < nhalo; ++c) {
        const nss_halo_t* halo = halos[c];
        NSS_REQUIRE(halo != nullptr, "p2p_create: NULL halo");
        NSS_REQUIRE(halo->direct || halo->n_send == 0, "p2p_create: the halos must send contiguous runs (direct)");
        nss_p2p_s::Channel ch;
        ch.key_send = halo->h_send_off;
        ch.key_recv = halo->h_recv_off;
        ch.n_owned = n_owned[c];
        ch.recv_off.assign(size_t(nranks), -1);
        ch.recv_cnt.assign(size_t(nranks), 0);
        int64_t total = 0;
        for (int i = 0; i < halo->n_recv; ++i) {
          const int q = halo->h_recv_peer[i];
          NSS_REQUIRE(q >= 0 && q < nranks && q != rank, "p2p_create: bad source rank");
          ch.recv_off[size_t(q)] = halo->h_recv_off[i] - n_owned[c];
          ch.recv_cnt[size_t(q)] = halo->h_recv_cnt[i];
          total += halo->h_recv_cnt[i];
        }
        ch.landing_doubles = std::max<int64_t>(total, 1);
        ch.flags_off = off;
        off += flags_bytes();
        p->channels.push_back(ch);
      }
      // landing zones behind all flag rows; their sizes differ from rank to rank, so every rank publishes ITS offsets
      // relative to a layout that only depends on (nranks, nhalo): zone c starts at a 256-byte aligned offset that the
      // OWNER computes -- the sender needs the owner's value, which travels in the blob
      for (int c = 0; c < nhalo; ++c) {
        off = (off + 255) & ~size_t(255);
        p->channels[size_t(c)].landing_off = off;
        p->channels[size_t(c)].zone_bytes = (sizeof(double) * size_t(p->channels[size_t(c)].landing_doubles) + 255) & ~size_t(255);
        off += 2 * p->channels[size_t(c)].zone_bytes;
      }
      p->region_bytes = off;
      // fine-grained device memory: remote stores become visible to this GPU's loads without cache maintenance
      void* mem = nullptr;
      NSS_HIP(hipExtMallocWithFlags(&mem, p->region_bytes, hipDeviceMallocFinegrained));
      p->region = static_cast<char*>(mem);
      NSS_HIP(hipMemset(p->region, 0, p->region_bytes));
      p->mail = reinterpret_cast<unsigned long long*>(p->region);
      NSS_HIP(hipMalloc(&p->d_error, sizeof(int32_t)));
      NSS_HIP(hipMemset(p->d_error, 0, sizeof(int32_t)));
      NSS_HIP(hipMalloc(&p->d_ticket, sizeof(int32_t) * kP2pMaxSegments));
      NSS_HIP(hipMemset(p->d_ticket, 0, sizeof(int32_t) * kP2pMaxSegments));
      NSS_HIP(hipDeviceSynchronize());
      hipIpcMemHandle_t handle;
      NSS_HIP(hipIpcGetMemHandle(&handle, p->region));
      static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
      char* blob = static_cast<char*>(h_blob);
      std::memcpy(blob, &handle, 64);
      for (int c = 0; c < nhalo; ++c) {
        // what the senders need: where (in bytes from the start of MY region) rank q's segment goes, and how long it is
        std::vector<int64_t> where(size_t(nranks), -1);
        for (int q = 0; q < nranks; ++q)
          if (p->channels[size_t(c)].recv_off[size_t(q)] >= 0)
            where[size_t(q)] = int64_t(p->channels[size_t(c)].landing_off) + 8 * p->channels[size_t(c)].recv_off[size_t(q)];
        char* at = blob + 64 + chan_blob(nranks) * c;
        std::memcpy(at, where.data(), sizeof(int64_t) * size_t(nranks));
        std::memcpy(at + 8 * size_t(nranks), p->channels[size_t(c)].recv_cnt.data(), sizeof(int64_t) * size_t(nranks));
        const int64_t zb = int64_t(p->channels[size_t(c)].zone_bytes);
        std::memcpy(at + 16 * size_t(nranks), &zb, sizeof(int64_t));
      }
    } catch (...) {
      nss_p2p_destroy(p);
      throw;
    }
    *out = p;
  });
}

int nss_p2p_connect(nss_p2p_t p, const void* h_blobs) {
  return guarded([&] {
    NSS_REQUIRE(p && h_blobs, "p2p_connect: NULL argument");
    NSS_REQUIRE(!p->connected, "p2p_connect: already connected");
    const int nchan = int(p->channels.size());
    const size_t blob = blob_bytes(p->nranks, nchan);
    const char* all = static_cast<const char*>(h_blobs);
    p->peer_region.assign(size_t(p->nranks), nullptr);
    for (nss_p2p_s::Channel& ch : p->channels) {
      ch.peer_land_off.assign(size_t(p->nranks), -1);
      ch.peer_zone_bytes.assign(size_t(p->nranks), 0);
    }
    std::vector<unsigned long long*> mails(size_t(p->nranks), nullptr);
    for (int q = 0; q < p->nranks; ++q) {
      if (q == p->rank) {
        p->peer_region[size_t(q)] = p->region;
      } else {
        hipIpcMemHandle_t handle;
        std::memcpy(&handle, all + blob * size_t(q), 64);
        void* mapped = nullptr;
        NSS_HIP(hipIpcOpenMemHandle(&mapped, handle, hipIpcMemLazyEnablePeerAccess));
        p->peer_region[size_t(q)] = static_cast<char*>(mapped);
        for (int c = 0; c < nchan; ++c) {
          int64_t where = -1;                              // byte offset inside q's region where q wants OUR segment
          const char* at = all + blob * size_t(q) + 64 + chan_blob(p->nranks) * c;
          std::memcpy(&where, at + 8 * size_t(p->rank), sizeof(int64_t));
          int64_t zb = 0;
          std::memcpy(&zb, at + 16 * size_t(p->nranks), sizeof(int64_t));
          p->channels[size_t(c)].peer_land_off[size_t(q)] = where;      // (the zones of different ranks start at different offsets)
          p->channels[size_t(c)].peer_zone_bytes[size_t(q)] = zb;
        }
      }
      mails[size_t(q)] = reinterpret_cast<unsigned long long*>(p->peer_region[size_t(q)]);
    }
    NSS_HIP(hipMalloc(&p->d_peer_mail, sizeof(unsigned long long*) * size_t(p->nranks)));
    NSS_HIP(hipMemcpy(p->d_peer_mail, mails.data(), sizeof(unsigned long long*) * size_t(p->nranks), hipMemcpyHostToDevice));
    p->connected = true;
  });
}

int nss_p2p_destroy(nss_p2p_t p) {
  return guarded([&] {
    if (!p) return;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < int(p->peer_region.size()); ++q)
      if (q != p->rank && p->peer_region[size_t(q)]) (void)hipIpcCloseMemHandle(p->peer_region[size_t(q)]);
    (void)hipFree(p->region);
    (void)hipFree(p->d_peer_mail);
    (void)hipFree(p->d_error);
    (void)hipFree(p->d_ticket);
    delete p;
  });
}

int nss_p2p_allreduce_f64(nss_p2p_t p, const double* src, double* dst, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(p && src && dst, "p2p_allreduce: bad argument");
    p2p_allreduce(*p, src, dst, as_stream(stream));
  });
}

int nss_p2p_exchange(nss_p2p_t p, const nss_halo_t* halo, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(p && halo, "p2p_exchange: NULL argument");
    p2p_exchange(*p, *halo, nullptr, as_stream(stream));
  });
}

int nss_p2p_error(nss_p2p_t p, int32_t* timed_out, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(p && timed_out, "p2p_error: NULL argument");
    NSS_HIP(hipMemcpyAsync(timed_out, p->d_error, sizeof(int32_t), hipMemcpyDeviceToHost, as_stream(stream)));
    NSS_HIP(hipStreamSynchronize(as_stream(stream)));
  });
}

}  // extern "C"

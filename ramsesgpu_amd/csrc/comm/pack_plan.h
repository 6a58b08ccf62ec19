// pack_plan.h -- where the chunks of one halo exchange go inside the staging buffers of the packed exchange (hip/rg_transport.h).
// Plain C++ (no HIP, no RCCL) so that tests/ can check it on the host for topologies a one-GPU box cannot run
// (tests/test_pack_plan.py: rings of 1, 2, 3, 5 slabs and open-ended stacks).
//
// Rule: the operations for ONE peer are concatenated IN POSTING ORDER -- sends into a region of the send stage, receives into a
// region of the receive stage -- and travel as one send and one receive per peer.  Grouped point-to-point operations between two
// ranks match in posting order, so the i-th chunk rank a sends to rank b is the i-th chunk b receives from a, packed or not:
// concatenation keeps every chunk where the in-place exchange would have put it.
#pragma once
#include <cstddef>

namespace rgpu_transport {

enum { kPackMaxOps = 64, kPackMaxPeers = 4, kPackMaxChunksPerDirection = 32 };

struct PackPlan {
  int npeers;
  int peer[kPackMaxPeers];                                   // distinct peers, in order of first appearance
  size_t send_base[kPackMaxPeers], send_total[kPackMaxPeers];   // region of the send stage per peer (doubles)
  size_t recv_base[kPackMaxPeers], recv_total[kPackMaxPeers];   // region of the receive stage per peer
  size_t off[kPackMaxOps];                                   // offset of operation i inside ITS stage (send or receive)
  size_t stage_doubles;                                      // capacity either stage needs
  size_t longest;                                            // longest chunk (launch geometry of the copy kernels)
  int nsend, nrecv;
};

// peer_of / count_of / is_send: the operation list.  Returns 0, or -1 (too many operations / peers / chunks).
template <class Ops>
inline int pack_plan(const Ops* ops, int nops, PackPlan* pl) {
  if (nops > kPackMaxOps) return -1;
  pl->npeers = 0; pl->stage_doubles = 0; pl->longest = 0; pl->nsend = 0; pl->nrecv = 0;
  int pidx[kPackMaxOps];
  for (int q = 0; q < kPackMaxPeers; ++q) { pl->send_total[q] = 0; pl->recv_total[q] = 0; pl->send_base[q] = 0; pl->recv_base[q] = 0; pl->peer[q] = -1; }
  for (int i = 0; i < nops; ++i) {
    int q = 0;
    while (q < pl->npeers && pl->peer[q] != ops[i].peer) ++q;
    if (q == pl->npeers) { if (pl->npeers == kPackMaxPeers) return -1; pl->peer[pl->npeers++] = ops[i].peer; }
    pidx[i] = q;
    (ops[i].send ? pl->send_total : pl->recv_total)[q] += ops[i].count;
    if (ops[i].send) ++pl->nsend; else ++pl->nrecv;
    if (ops[i].count > pl->longest) pl->longest = ops[i].count;
  }
  if (pl->nsend > kPackMaxChunksPerDirection || pl->nrecv > kPackMaxChunksPerDirection) return -1;
  size_t all_s = 0, all_r = 0;
  for (int q = 0; q < pl->npeers; ++q) { pl->send_base[q] = all_s; all_s += pl->send_total[q]; pl->recv_base[q] = all_r; all_r += pl->recv_total[q]; }
  pl->stage_doubles = all_s > all_r ? all_s : all_r;
  size_t fill_s[kPackMaxPeers] = {0, 0, 0, 0}, fill_r[kPackMaxPeers] = {0, 0, 0, 0};
  for (int i = 0; i < nops; ++i) {
    const int q = pidx[i];
    if (ops[i].send) { pl->off[i] = pl->send_base[q] + fill_s[q]; fill_s[q] += ops[i].count; }
    else { pl->off[i] = pl->recv_base[q] + fill_r[q]; fill_r[q] += ops[i].count; }
  }
  return 0;
}

}  // namespace rgpu_transport

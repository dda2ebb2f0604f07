// wf_groups.hip — direction groups: farms partitioned by a small set of distinct wind directions (the rows of a shared
// wind series, the grid directions of binned reset sampling), one sorted geometry + pair table per group.
#include <algorithm>

#include "wf_handle.h"

namespace wfi {

// leaving a grouped launch: back to the choice for the plain batch
void ungroup(wf_handle* h) {
  if (h->n_groups > 0 && h->ll_G) {
    const int llg = pick_ll(h, h->N, h->B);
    if (llg && ((llg >> 4) != h->ll_G || (llg & 15) != h->ll_S)) {
      hipStreamSynchronize(h->stream);
      set_ll_shape(h, llg >> 4, llg & 15);
      reset_calibration(h);  // (back on the rounds model's guess: the plain batch is timed again on its third step)
    }
  }
  h->n_groups = 0;
}
// Farms per block of the table-path launch of the handle's kernel variant (wf_step_kernel), and of the
// one-block-at-a-time kernel when it is in use.  A grouped launch pads every group to a multiple of the larger of the
// two (both are powers of two), and its block -> group list has one entry per `group_unit` farms (the smaller).
int farms_per_block(const wf_handle* h) {
  int vG, vS; const void* vfn;
  wfk_variant(h->variant, &vG, &vS, &vfn);
  return wfk_tab_waves() * (64 / vG);
}
int group_pad(const wf_handle* h) {
  const int a = farms_per_block(h), b = h->ll_G ? wfk_ll_farms_per_block(h->ll_G) : 0;
  return a > b ? a : b;
}
int group_unit(const wf_handle* h) {
  const int a = farms_per_block(h), b = h->ll_G ? wfk_ll_farms_per_block(h->ll_G) : a;
  return a < b ? a : b;
}

// Would a grouped launch over K direction groups pay off?  Every group is padded to whole blocks (half a block wasted
// per group on average) against the ~2x cost of the on-the-fly path.
bool groups_pay_off(const wf_handle* h, int K) { return h->n_layouts == 1 && groups_fit(h, K); }
// ... the same question for any partition of the farms into K groups with a geometry each (direction groups of one
// layout; the layouts of a batch under one direction: wf_set_layouts)
bool groups_fit(const wf_handle* h, int K) {
  if (h->N > WF_PAIR_MAX_N || !wfk_variant_has_table(h->variant) || h->choice.pair_table == 0 || K < 1) return false;
  if ((size_t)K * h->N > h->cap_bn) return false;  // group geometry lives in the per-farm geometry buffers
  // (priced at the 64-farm blocks of G = 4: build_groups leaves the G = 2 kernel unless its 128-farm blocks are cheaper)
  const int pad = std::max(farms_per_block(h), h->ll_G ? wfk_ll_farms_per_block(h->ll_G == 2 ? 4 : h->ll_G) : 0);
  const double waste = 0.5 * pad * K / (double)h->B;
  int vG, vS; const void* vfn;
  wfk_variant(h->variant, &vG, &vS, &vfn);
  const size_t bytes = (size_t)K * h->N * WF_PAIR_ROW_FLOATS(vG * vS) * sizeof(float);
  return waste < 0.5 && bytes <= ((size_t)8 << 30);
}

// Partition the farms by `group_of_farm` (host, B entries in [0, K)): farm list sorted by group and padded per group to
// whole blocks (d_perm, -1 = padding), group of each block (d_blk_group).  Then the sorted geometry of the K
// directions `d_wd_groups` (device) is built into the geometry buffers; the pair tables follow lazily (pair_table()).
int build_groups(wf_handle* h, const int* group_of_farm, int K, const double* d_wd_groups, bool rebuild_geometry) {
  std::vector<int> count(K, 0);
  for (int b = 0; b < h->B; ++b) {
    if (group_of_farm[b] < 0 || group_of_farm[b] >= K) return fail(h, WF_E_INVALID, "direction group out of range");
    ++count[group_of_farm[b]];
  }
  // The 128-farm blocks of the G = 2 kernel double the padding of every group: it stays only where the padded launch
  // is still its cheapest (few large groups that divide into its blocks: 65 536 farms over two layouts); otherwise
  // G = 4 (the choice between its two kernels follows the padded count, below).  (Also when the choice forces G = 2
  // for the plain batch: the lists are laid out for the block size decided here.)
  if (h->ll_G == 2) {
    auto padded = [&](int pad) {
      long n = 0;
      for (int g = 0; g < K; ++g) n += (long)((count[g] + pad - 1) / pad) * pad;
      return n;
    };
    const int fs = farms_per_block(h), p2 = std::max(fs, wfk_ll_farms_per_block(2)), p4 = std::max(fs, wfk_ll_farms_per_block(4));
    const double t2 = ll_estimate(h, 4, h->N, padded(p2));
    const double t4 = std::min(ll_estimate(h, 2, h->N, padded(p4)), ll_estimate(h, 3, h->N, padded(p4)));
    if (!(t2 < t4)) {
      WF_HIP(h, hipStreamSynchronize(h->stream));
      set_ll_shape(h, 4, 2);
    }
  }
  const int epb = group_pad(h), unit = group_unit(h);
  std::vector<int> first_slot(K, 0), blk_group;
  int slots = 0;
  for (int g = 0; g < K; ++g) {
    first_slot[g] = slots;
    const int nb = (count[g] + epb - 1) / epb;
    for (int q = 0; q < nb * (epb / unit); ++q) blk_group.push_back(g);
    slots += nb * epb;
  }
  std::vector<int> perm(slots > 0 ? slots : 1, -1), cursor(first_slot);
  for (int b = 0; b < h->B; ++b) perm[cursor[group_of_farm[b]]++] = b;
  if (perm.size() > h->perm_cap) {
    hipFree(h->d_perm); h->d_perm = nullptr; h->perm_cap = 0;
    WF_HIP(h, hipMalloc(&h->d_perm, sizeof(int) * perm.size()));
    h->perm_cap = perm.size();
  }
  if (blk_group.size() > h->blk_cap) {
    hipFree(h->d_blk_group); h->d_blk_group = nullptr; h->blk_cap = 0;
    WF_HIP(h, hipMalloc(&h->d_blk_group, sizeof(int) * blk_group.size()));
    h->blk_cap = blk_group.size();
  }
  WF_HIP(h, hipStreamSynchronize(h->stream));  // a launch in flight may still read the previous lists
  WF_HIP(h, hipMemcpy(h->d_perm, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice));
  WF_HIP(h, hipMemcpy(h->d_blk_group, blk_group.data(), sizeof(int) * blk_group.size(), hipMemcpyHostToDevice));
  h->n_blocks = (int)blk_group.size();
  h->n_slots = slots;
  h->n_groups = K;
  h->group_shift = 0;
  {
    const int s_new = repick_ll_slots(h, h->N, h->ll_G, h->ll_S, (long)slots);
    if (s_new != h->ll_S) set_ll_shape(h, h->ll_G, s_new);
  }
  if (rebuild_geometry) {
    // group g: direction g of the one layout, or layout g under its direction (wf_set_layouts: groups are the layouts)
    WF_HIP(h, wfk_launch_geometry(K, h->N, h->d_lx, h->d_ly, h->d_centre, h->n_layouts == 1 ? 0 : 1, nullptr, h->d_layout_n, d_wd_groups, 1, h->d_gx, h->d_gy,
                                  h->d_gidx, 0, nullptr, nullptr, h->stream));
    h->pair_dirty = true;
  }
  return WF_OK;
}

}  // namespace wfi

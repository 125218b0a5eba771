/* GpuTopKPostingSource — seam B2 of SURVEY.md §8(b): the reference's own operator plug-in API
 * (Xapian::PostingSource, reference src/xapian/postingsource.h:76-390; adapted into the PostList
 * tree by ExternalPostList, src/xapian/matcher/externalpostlist.cc:40-183) in front of libxgm.so.
 *
 * A maintainer adds this file pair to Xapiand and wraps an eligible query in it:
 *
 *     auto* src = new GpuTopKPostingSource(idx, desc);       // desc: the AND / OR / PHRASE of terms
 *     Xapian::Query q(src->release());                       // Xapian owns it (postingsource.h:399-413)
 *     enquire.set_query(q);  mset = enquire.get_mset(first, maxitems);
 *
 * init() runs the whole query on the device (xgm_plan_query + xgm_search for first+maxitems hits) and
 * the source then replays those hits in docid order with their BM25 weights; Xapian's own ProtoMSet
 * re-establishes rank order (weight descending, docid ascending), so the MSet's docids, weights and
 * ranks are those of the CPU matcher.  ExternalPostList multiplies by `factor` (externalpostlist.cc:
 * 95-103): use the query unscaled (factor 1.0).  Not compiled into libxgm.so — it needs the host's
 * Xapian headers; oracle/ref_build/hook_driver.cc builds it against the reference for the parity test.
 */
#ifndef XGM_POSTING_SOURCE_H
#define XGM_POSTING_SOURCE_H

#include <xapian.h>

#include <string>
#include <vector>

#include "xgm.h"

class GpuTopKPostingSource : public Xapian::PostingSource {
  public:
    /* op: XGM_OP_AND / XGM_OP_OR / XGM_OP_PHRASE; terms in query order; k = first + maxitems the caller
     * will ask get_mset for; window: PHRASE only (0 = exact phrase). */
    GpuTopKPostingSource(xgm_index* idx, uint32_t op, const std::vector<std::string>& terms, uint32_t k, uint32_t window = 0);

    Xapian::doccount get_termfreq_min() const override { return matches_; }
    Xapian::doccount get_termfreq_est() const override { return matches_; }
    Xapian::doccount get_termfreq_max() const override { return matches_; }
    double get_weight() const override { return by_docid_[pos_].weight; }
    Xapian::docid get_docid() const override { return by_docid_[pos_].docid; }
    void next(double min_wt) override;
    void skip_to(Xapian::docid did, double min_wt) override;
    bool at_end() const override { return started_ && pos_ >= by_docid_.size(); }
    PostingSource* clone() const override { return new GpuTopKPostingSource(idx_, op_, terms_, k_, window_); }
    std::string name() const override { return "GpuTopKPostingSource"; }
    void init(const Xapian::Database& db) override;
    std::string get_description() const override { return "GpuTopKPostingSource(libxgm)"; }

    /* XGM_OK after init(); > 0: the device path declined the query (run the CPU matcher instead) */
    int status() const { return status_; }

  private:
    xgm_index* idx_;
    uint32_t op_;
    std::vector<std::string> terms_;
    uint32_t k_, window_;
    std::vector<xgm_hit> by_docid_;
    size_t pos_ = 0;
    bool started_ = false;
    Xapian::doccount matches_ = 0;
    int status_ = XGM_OK;
};

#endif

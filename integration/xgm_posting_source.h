/* GpuTopKPostingSource — seam B2 of SURVEY.md §8(b): the reference's own operator plug-in API
 * (Xapian::PostingSource, reference src/xapian/postingsource.h:76-390; adapted into the PostList
 * tree by ExternalPostList, src/xapian/matcher/externalpostlist.cc:40-183) in front of libxgm.so.
 *
 * A maintainer adds this file pair to Xapiand and wraps an eligible query in it:
 *
 *     auto* src = GpuTopKPostingSource::create(idx, op, terms, k, window, &merged_stats);   // nullptr: not eligible,
 *     if (src) query = Xapian::Query(src->release());         // keep the original Xapian::Query and the CPU matcher
 *     enquire.set_query(query);  mset = enquire.get_mset(first, maxitems);
 *
 * create() PLANS the query (xgm_plan_query) before the Xapian::Query is replaced, so a shape the device path
 * declines never reaches init(); the constructor form below stays for callers that know the shape is eligible and
 * throws Xapian::UnimplementedError from init() if it is not (it never degrades to an empty result).  On a
 * multi-shard index pass the MERGED statistics (what Enquire::add_prepared_mset accumulated, enquire.cc:385-394):
 * Xapiand's per-shard get_mset weighs with them, and so must the plug-in.
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
    GpuTopKPostingSource(xgm_index* idx, uint32_t op, const std::vector<std::string>& terms, uint32_t k, uint32_t window = 0,
                         const xgm_global_stats* merged = nullptr);

    /* Eligibility step: plans the query now; nullptr (and *status = the xgm return code, when given) if the device path
     * declines it or the plan fails — the caller keeps its Xapian::Query. */
    static GpuTopKPostingSource* create(xgm_index* idx, uint32_t op, const std::vector<std::string>& terms, uint32_t k, uint32_t window = 0,
                                        const xgm_global_stats* merged = nullptr, int* status = nullptr);

    /* at most k documents are replayed: the lower bound is what init() found among them, not the exact match count */
    Xapian::doccount get_termfreq_min() const override { return (Xapian::doccount)by_docid_.size(); }
    Xapian::doccount get_termfreq_est() const override { return matches_; }
    Xapian::doccount get_termfreq_max() const override { return matches_; }
    double get_weight() const override { return by_docid_[pos_].weight; }
    Xapian::docid get_docid() const override { return by_docid_[pos_].docid; }
    void next(double min_wt) override;
    void skip_to(Xapian::docid did, double min_wt) override;
    bool at_end() const override { return started_ && pos_ >= by_docid_.size(); }
    PostingSource* clone() const override { return new GpuTopKPostingSource(idx_, op_, terms_, k_, window_, have_merged_ ? &merged_ : nullptr); }
    std::string name() const override { return "GpuTopKPostingSource"; }
    void init(const Xapian::Database& db) override;
    std::string get_description() const override { return "GpuTopKPostingSource(libxgm)"; }

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
    bool have_merged_ = false;
    xgm_global_stats merged_;

    void describe(xgm_query_desc* d) const;
};

#endif

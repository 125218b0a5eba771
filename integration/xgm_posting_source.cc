/* See xgm_posting_source.h. */
#include "xgm_posting_source.h"

#include <algorithm>

#include <cstring>

GpuTopKPostingSource::GpuTopKPostingSource(xgm_index* idx, uint32_t op, const std::vector<std::string>& terms, uint32_t k, uint32_t window,
                                           const xgm_global_stats* merged)
    : idx_(idx), op_(op), terms_(terms), k_(k), window_(window), have_merged_(merged != nullptr) {
    memset(&merged_, 0, sizeof merged_);
    if (merged) merged_ = *merged;
}

void GpuTopKPostingSource::describe(xgm_query_desc* d) const {
    memset(d, 0, sizeof *d);
    d->op = op_;
    d->n_terms = (uint32_t)terms_.size();
    for (size_t i = 0; i < terms_.size() && i < XGM_MAX_TERMS; ++i) { d->terms[i] = terms_[i].data(); d->term_len[i] = (uint32_t)terms_[i].size(); }
    d->window = window_;
    d->first = 0; d->maxitems = k_; d->check_at_least = 0;
    d->k1 = 1; d->k2 = 0; d->k3 = 1; d->b = 0.5; d->min_normlen = 0.5;        /* BM25Weight defaults, weight.h:635-667 */
}

GpuTopKPostingSource* GpuTopKPostingSource::create(xgm_index* idx, uint32_t op, const std::vector<std::string>& terms, uint32_t k, uint32_t window,
                                                   const xgm_global_stats* merged, int* status) {
    int rc = XGM_UNSUPPORTED;
    if (idx && !terms.empty() && terms.size() <= XGM_MAX_TERMS) {
        GpuTopKPostingSource probe(idx, op, terms, k, window, merged);
        xgm_query_desc d;
        probe.describe(&d);
        xgm_query q;
        rc = xgm_plan_query(idx, &d, merged, &q);
    }
    if (status) *status = rc;
    return rc == XGM_OK ? new GpuTopKPostingSource(idx, op, terms, k, window, merged) : nullptr;
}

void GpuTopKPostingSource::init(const Xapian::Database&) {
    by_docid_.clear();
    pos_ = 0;
    started_ = false;
    matches_ = 0;
    /* a declined query must never look like "no results": the caller was supposed to check with create() */
    if (terms_.empty() || terms_.size() > XGM_MAX_TERMS) { status_ = XGM_UNSUPPORTED; throw Xapian::UnimplementedError("GpuTopKPostingSource: query shape not handled by the device path"); }
    xgm_query_desc d;
    describe(&d);
    xgm_query q;
    status_ = xgm_plan_query(idx_, &d, have_merged_ ? &merged_ : nullptr, &q);
    if (status_ < 0) throw Xapian::DatabaseError(xgm_last_error());
    if (status_ > 0) throw Xapian::UnimplementedError("GpuTopKPostingSource: query shape not handled by the device path");
    std::vector<xgm_hit> hits(std::max<uint32_t>(1u, q.first + q.maxitems));
    xgm_result_hdr h;
    status_ = xgm_search(idx_, &q, hits.data(), &h);
    if (status_ < 0) throw Xapian::DatabaseError(xgm_last_error());
    if (status_ > 0) throw Xapian::UnimplementedError("GpuTopKPostingSource: the device path declined the batch");
    hits.resize(h.n_hits);
    std::sort(hits.begin(), hits.end(), [](const xgm_hit& a, const xgm_hit& b) { return a.docid < b.docid; });
    by_docid_.swap(hits);
    matches_ = (Xapian::doccount)XGM_MATCHES_COUNT(h.matches_exact);
    set_maxweight(h.max_possible);
}

void GpuTopKPostingSource::next(double) {
    if (!started_) { started_ = true; pos_ = 0; } else if (pos_ < by_docid_.size()) ++pos_;
}

void GpuTopKPostingSource::skip_to(Xapian::docid did, double) {
    if (!started_) { started_ = true; pos_ = 0; }
    while (pos_ < by_docid_.size() && by_docid_[pos_].docid < did) ++pos_;
}

// ema_amd/csrc/opts.h -- engine options -> kernel options (host side).
#ifndef EMA_OPTS_H
#define EMA_OPTS_H
#include <cstdlib>
#include "dev_types.h"
#include "../../include/ema_engine.h"

// bwa_fill_scmat(a, b, mat) + the derived constants of mem_collect_intv
inline DevOpts ema_make_dev_opts(const ema_engine_opts &o)
{
	DevOpts d;
	d.a = o.a; d.b = o.b; d.o_del = o.o_del; d.e_del = o.e_del; d.o_ins = o.o_ins; d.e_ins = o.e_ins;
	d.pen_clip5 = o.pen_clip5; d.pen_clip3 = o.pen_clip3; d.w = o.w; d.zdrop = o.zdrop;
	d.min_seed_len = o.min_seed_len;
	d.split_len = (int)(o.min_seed_len * o.split_factor + .499);
	d.split_width = o.split_width; d.max_mem_intv = o.max_mem_intv; d.max_occ = o.max_occ;
	d.max_chain_gap = o.max_chain_gap; d.min_chain_weight = o.min_chain_weight; d.max_chain_extend = o.max_chain_extend;
	d.mask_level = o.mask_level; d.drop_ratio = o.drop_ratio; d.mask_level_redun = o.mask_level_redun;
	d.intv_cap = EMA_INTV_CAP; d.reg_cap = EMA_REG_CAP; d.cig_cap = EMA_CIG_CAP; d.seed_budget = 1 << 30;
	{   // (ema_tuning_get: engine.hip; each call's result is consumed before the next)
		const char *v = ema_tuning_get("seed_wtest"); const int f0 = (v && atoi(v) == 0) ? 0 : 1;
		v = ema_tuning_get("seed_anchor"); const int f1 = (v && atoi(v) == 0) ? 0 : 2;
		v = ema_tuning_get("seed_onepass"); const int f2 = (v && atoi(v) == 0) ? 0 : 4;
		d.seed_flags = f0 | f1 | f2;
	}
	d.seed_ext = nullptr;      // (bit 3 of seed_flags and this array: set per slice by the engine, engine.hip slice_alloc)      // (the switch exists for A/B runs and for the parity tests of both routes)
	int k = 0;
	for (int i = 0; i < 4; ++i) {
		for (int j = 0; j < 4; ++j) d.mat[k++] = (int8_t)(i == j ? o.a : -o.b);
		d.mat[k++] = -1;
	}
	for (int j = 0; j < 5; ++j) d.mat[k++] = -1;
	return d;
}

inline void ema_fill_default_opts(ema_engine_opts *o)
{
	o->a = 1; o->b = 4; o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1;
	o->pen_clip5 = o->pen_clip3 = 5; o->w = 100; o->zdrop = 100;
	o->min_seed_len = 19; o->split_width = 10; o->max_mem_intv = 20;
	o->max_occ = 3000;            // reference src/align.c:185
	o->max_chain_gap = 10000; o->min_chain_weight = 0; o->max_chain_extend = 1 << 30;
	o->split_factor = 1.5f; o->mask_level = 0.50f; o->drop_ratio = 0.50f; o->mask_level_redun = 0.95f;
	o->score_delta = 25; o->max_rescue = 50; o->pes_low = -35; o->pes_high = 500;
	o->batch_pairs = 0; o->n_streams = 0; o->full_tier_pairs = 0;
	o->lean_intervals = o->lean_regions = o->lean_cigar_ops = 0; o->lean_seed_extends = 0;
	o->mapq_coef_len = 50; o->mapq_coef_fac = 3;
}
#endif

// ema_amd/csrc/host_index.h -- reads the on-disk index and lays it out for HBM.
// Replaces bwa_idx_load() as called by the reference's load_reference()
// (reference src/bwabridge.c:77-96).  Pure host C++, no HIP calls.
#ifndef EMA_HOST_INDEX_H
#define EMA_HOST_INDEX_H

#include <cstdint>
#include <string>
#include <vector>
#include "dev_types.h"

struct HostContig { std::string name; int64_t offset; int32_t len; int32_t is_alt; };

struct HostIndex {
	std::vector<OccBlock> occ;       // device layout (see dev_types.h)
	uint64_t occ_super[EMA_OCC_MAX_SUPER - 1][4] = {};
	int n_super = 1;
	std::vector<uint8_t> sa_bytes;   // seq_len+1 rows of sa_width bytes (empty when loaded without the suffix array)
	std::string sa_path;             // where those rows are on disk: sa_size bytes from sa_file_off
	uint64_t sa_file_off = 0, sa_size = 0;
	// A stock bwa index has no flat suffix array (<prefix>.fsa is this repo's builder's), only bwa's sampled one: <prefix>.sa holds
	// SA[j * sa_intv] for j = 1 .. seq_len / sa_intv.  Then sa_path is empty, sa_sampled = those values with entry 0 = -1 (bwa's
	// bwt_restore_sa leaves sa[0] = -1), and the rows in between follow by LF-mapping (bwt_sa): host_expand_sa below, or the
	// engine's kernel on the device (k_kmer.hip, ema_k_sa_expand).
	std::vector<uint64_t> sa_sampled;
	int sa_intv = 0;
	std::vector<uint8_t> pac;
	std::vector<int64_t> ctg_off;    // n+1
	std::vector<uint8_t> ctg_alt;    // n flags from <prefix>.alt; empty when no contig is ALT
	std::vector<int32_t> ctg_tab;    // dev_types.h, DevIndex::ctg_tab
	int ctg_shift = 0;
	std::vector<HostContig> contigs;
	uint64_t primary = 0, seq_len = 0, L2[5] = {0, 0, 0, 0, 0};
	int64_t l_pac = 0;
	int sa_width = 0;
	// k-mer interval table (dev_types.h), when a host copy exists (the SIMT harness of tests/ builds one; the engine builds its
	// own on the device)
	std::vector<uint64_t> kmer_wide, kmer_narrow;
	int kmer_k = 0;
	std::vector<uint64_t> text2;     // DevIndex::text2, when a host copy exists (the SIMT harness)
	// Fills a DevIndex whose pointers refer to THIS object's host buffers
	// (used by the host-side SIMT harness in tests/; the engine overwrites the
	// pointers with device addresses after upload).
	DevIndex view() const;
};

void host_contig_table(const std::vector<int64_t> &ctg_off, std::vector<int32_t> &tab, int &shift);

// Loads <prefix>.bwt/.pac/.ann and <prefix>.fsa -- or, without an .fsa, bwa's own <prefix>.sa (a stock `bwa index`).  Returns "" on success, else an error message.  with_sa == false
// leaves the suffix array on disk (sa_path / sa_file_off / sa_size say where): the engine streams it to the device.
std::string host_index_load(const std::string &prefix, HostIndex &out, bool with_sa = true);

// The flat suffix array (ix.sa_bytes: seq_len + 1 rows of ix.sa_width bytes) from bwa's sampled one and the rank structure, as
// bwt_sa() computes single rows (reference path: src/bwabridge.c:79 bwa_idx_load -> bwt_restore_sa; rows located in mem_chain).
void host_expand_sa(HostIndex &ix);

#endif

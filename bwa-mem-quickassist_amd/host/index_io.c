/*
 * index_io.c -- reads the index files `bwa index` writes, without the reference's code, into the plain arrays the
 * C-ABI takes (bmh_bwt_t for bmh_ctx_set_bwt, the 2-bit reference for bmh_ctx_set_pac):
 *   <prefix>.bwt  u64 primary, u64 L2[1..4], then the BWT words to the end of the file -- already in the layout with
 *                 the occurrence counts interleaved (bwt_restore_bwt, reference bwa-0.7.8/bwt.c:403-421)
 *   <prefix>.sa   u64 primary, 4 x u64 (L2 again), u64 sa_intv, u64 seq_len, then sa[1..n_sa); sa[0] = -1
 *                 (bwt_restore_sa, bwt.c:380-401)
 *   <prefix>.ann  "l_pac n_seqs seed", then per sequence "gi name [comment]" and "offset len n_ambs"
 *                 (bns_restore_core, bntseq.c:94-140)
 *   <prefix>.amb  "l_pac n_seqs n_holes", then per run of ambiguous bases "offset len letter" (bntseq.c:136-150); the
 *                 2-bit reference holds a random base there
 *   <prefix>.pac  l_pac/4+1 bytes of 2-bit codes, four per byte, first base in the top bits (bwa.c:291-292)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bwamem_hip.h"

static FILE *open_ext(const char *prefix, const char *ext, const char *mode)
{
	char *fn = (char *)malloc(strlen(prefix) + strlen(ext) + 1);
	FILE *f;
	strcat(strcpy(fn, prefix), ext);
	f = fopen(fn, mode);
	free(fn);
	return f;
}

static int read_all(FILE *f, void *dst, size_t bytes)
{
	size_t off = 0;
	while (off < bytes) {
		const size_t want = bytes - off < ((size_t)16 << 20) ? bytes - off : (size_t)16 << 20, got = fread((char *)dst + off, 1, want, f);
		if (got == 0) return -1;
		off += got;
	}
	return 0;
}

void bmh_index_free(bmh_index_t *ix)
{
	int i;
	if (!ix) return;
	free((void *)ix->bwt.bwt), free((void *)ix->bwt.sa), free(ix->pac);
	for (i = 0; i < ix->n_seqs; ++i)
		if (ix->names) free(ix->names[i]);
	free(ix->names), free(ix->offsets), free(ix->lens), free(ix->hole_offsets), free(ix->hole_lens), free(ix->hole_chars), free(ix);
}

int bmh_index_load(const char *prefix, bmh_index_t **out)
{
	bmh_index_t *ix;
	FILE *f = 0;
	uint64_t hdr[7];
	long end;
	int i;
	if (!prefix || !out) return BMH_E_ARG;
	*out = 0;
	if (!(ix = (bmh_index_t *)calloc(1, sizeof(*ix)))) return BMH_E_NOMEM;

	if (!(f = open_ext(prefix, ".bwt", "rb"))) goto fail;
	if (fseek(f, 0, SEEK_END) || (end = ftell(f)) < 40 || fseek(f, 0, SEEK_SET)) goto fail;
	ix->bwt.bwt_size = ((uint64_t)end - 40) >> 2;
	if (read_all(f, hdr, 40)) goto fail;
	ix->bwt.primary = hdr[0], ix->bwt.L2[0] = 0;
	for (i = 1; i < 5; ++i) ix->bwt.L2[i] = hdr[i];
	ix->bwt.seq_len = ix->bwt.L2[4];
	if (!(ix->bwt.bwt = (const uint32_t *)malloc((size_t)ix->bwt.bwt_size * 4 + 64))) goto fail;
	if (read_all(f, (void *)ix->bwt.bwt, (size_t)ix->bwt.bwt_size * 4)) goto fail;
	fclose(f);

	if (!(f = open_ext(prefix, ".sa", "rb"))) goto fail;
	if (read_all(f, hdr, 56) || hdr[0] != ix->bwt.primary || hdr[6] != ix->bwt.seq_len || hdr[5] < 1 || (hdr[5] & (hdr[5] - 1))) goto fail;
	ix->bwt.sa_intv = (int32_t)hdr[5];
	ix->bwt.n_sa = (ix->bwt.seq_len + hdr[5]) / hdr[5];
	if (!(ix->bwt.sa = (const uint64_t *)malloc((size_t)ix->bwt.n_sa * 8 + 64))) goto fail;
	((uint64_t *)ix->bwt.sa)[0] = (uint64_t)-1;
	if (read_all(f, (uint64_t *)ix->bwt.sa + 1, (size_t)(ix->bwt.n_sa - 1) * 8)) goto fail;
	fclose(f);

	if (!(f = open_ext(prefix, ".ann", "r"))) goto fail;
	{
		long long l_pac;
		unsigned seed;
		char line[4096];
		if (fscanf(f, "%lld%d%u", &l_pac, &ix->n_seqs, &seed) != 3 || l_pac <= 0 || ix->n_seqs < 0) goto fail;
		ix->l_pac = l_pac;
		ix->names = (char **)calloc((size_t)ix->n_seqs + 1, sizeof(char *));
		ix->offsets = (int64_t *)calloc((size_t)ix->n_seqs + 1, 8), ix->lens = (int32_t *)calloc((size_t)ix->n_seqs + 1, 4);
		for (i = 0; i < ix->n_seqs; ++i) {
			long long off;
			int gi, len, n_ambs;
			char name[1024];
			if (fscanf(f, "%d%1023s", &gi, name) != 2) goto fail;
			if (!fgets(line, sizeof(line), f)) goto fail; /* the rest of the line is the optional comment */
			if (fscanf(f, "%lld%d%d", &off, &len, &n_ambs) != 3) goto fail;
			ix->names[i] = strdup(name), ix->offsets[i] = off, ix->lens[i] = len;
		}
	}
	fclose(f);

	if (!(f = open_ext(prefix, ".amb", "r"))) goto fail;
	{
		long long l_pac, off;
		int n_seqs, len;
		char letter[64];
		if (fscanf(f, "%lld%d%d", &l_pac, &n_seqs, &ix->n_holes) != 3 || l_pac != ix->l_pac || n_seqs != ix->n_seqs || ix->n_holes < 0) goto fail;
		ix->hole_offsets = (int64_t *)calloc((size_t)ix->n_holes + 1, 8), ix->hole_lens = (int32_t *)calloc((size_t)ix->n_holes + 1, 4);
		ix->hole_chars = (char *)calloc((size_t)ix->n_holes + 1, 1);
		for (i = 0; i < ix->n_holes; ++i) {
			if (fscanf(f, "%lld%d%63s", &off, &len, letter) != 3) goto fail;
			ix->hole_offsets[i] = off, ix->hole_lens[i] = len, ix->hole_chars[i] = letter[0];
		}
	}
	fclose(f);

	if (!(f = open_ext(prefix, ".pac", "rb"))) goto fail;
	if (!(ix->pac = (uint8_t *)calloc((size_t)(ix->l_pac / 4 + 1) + 16, 1))) goto fail;
	if (read_all(f, ix->pac, (size_t)(ix->l_pac / 4 + 1))) goto fail;
	fclose(f);
	if (ix->bwt.seq_len != (uint64_t)ix->l_pac * 2) goto fail_nofile; /* the BWT is over both strands */
	*out = ix;
	return BMH_OK;
fail:
	if (f) fclose(f);
fail_nofile:
	bmh_index_free(ix);
	return BMH_E_ARG;
}

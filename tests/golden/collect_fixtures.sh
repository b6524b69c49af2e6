#!/bin/sh
# Copies the DATA fixtures (inputs / expected-output files) that the reference's
# own tests use for this path from a reference checkout into tests/golden/.
# Data only: no reference source code is copied.  Run once in the authoring
# container; the GPU box only sees the committed copies.
set -e
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
D=$REF/tests/data
mkdir -p "$HERE/tokenizers" "$HERE/consensus" "$HERE/fragments/region_scoring" \
         "$HERE/fragments/fragsplit" "$HERE/out" "$HERE/regionset"
for f in peaks.bed peaks.bed.gz peaks.scored.bed peaks.scored.sorted.bed tokenizer.toml \
         tokenizer_ailist.toml tokenizer_bad_ttype.toml tokenizer_bits.toml \
         tokenizer_custom_specials.toml tokenizer_ordered.toml; do
  cp "$D/tokenizers/$f" "$HERE/tokenizers/$f"
done
cp "$D/to_tokenize.bed" "$HERE/"
cp "$REF/tests/hg38.chrom.sizes" "$HERE/"
cp "$D/consensus/consensus1.bed" "$HERE/consensus/"
cp "$D"/fragments/region_scoring/*.bed.gz "$HERE/fragments/region_scoring/"
cp "$D"/fragments/fragsplit/*.bed.gz "$HERE/fragments/fragsplit/"
cp "$D/barcode_cluster_map.tsv" "$HERE/"
cp -r "$D/igd_file_list_01" "$D/igd_file_list_02" "$D/igd_query_files" "$D/lola_multi_db" "$HERE/"
cp "$D/out/peaks.gtok" "$D/out/tokens.gtok" "$HERE/out/"
cp "$D/regionset/dummy.bed" "$D/regionset/dummy_b.bed" "$D/regionset/dummy_headers.bed" \
   "$D/regionset/dummy_incorrect_headers.bed" "$D/regionset/dummy.narrowPeak" \
   "$D/regionset/dummy.narrowPeak.bed.gz" "$HERE/regionset/"
cp "$D/test_sorted_small.bed" "$D/test_unsorted_small.bed" "$D/test_unknown_chrom.bed" "$HERE/"
chmod -R u+w "$HERE"

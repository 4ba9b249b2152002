cd $GRAFT_REPO_ROOT
# tools/k4chunk.sh — GPU box: k_inquad by chunk size (tiles of 1024 points per block) on the tuning build (tools/mkvariant.sh tuning "" -DSSD_TUNING: SSD_K4_CHUNK_TILES
# is read from the environment), tools/stages.py alternately, XGA then FHD stress; profiles/r06_k4_edges.txt
for r in 1 2; do
for t in 8 16 32 64 128; do
  SSD_K4_CHUNK_TILES=$t SSD_HIP_LIB=$GRAFT_REPO_ROOT/stair-step-detector_amd/lib_tuning/libssd_hip.so STAGES_TAG="k4 tiles $t" python tools/stages.py 1024 8
done; done
for r in 1 2; do
for t in 8 16 32 64; do
  STAGES_FHD=1 SSD_K4_CHUNK_TILES=$t SSD_HIP_LIB=$GRAFT_REPO_ROOT/stair-step-detector_amd/lib_tuning/libssd_hip.so STAGES_TAG="fhd k4 tiles $t" python tools/stages.py 256 8
done; done

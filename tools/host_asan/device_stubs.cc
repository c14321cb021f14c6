// (2 of 2: no header here -- the stand-ins take any arguments.)
extern "C" int dv_host_asan_no_device(const char* name);
#define DV_STUB(name) extern "C" int name(...) { return dv_host_asan_no_device(#name); }

DV_STUB(dv_crc32c) DV_STUB(dv_downsample_indices) DV_STUB(dv_last_profile_count) DV_STUB(dv_profile_ms)
DV_STUB(dv_query_reads) DV_STUB(dv_set_profiling)
DV_STUB(dv_base_aux_plane) DV_STUB(dv_encode_batch) DV_STUB(dv_encoder_create) DV_STUB(dv_encoder_destroy)
DV_STUB(dv_validate_batch)
DV_STUB(dv_model_conv_macs) DV_STUB(dv_model_create) DV_STUB(dv_model_debug_tensor) DV_STUB(dv_model_destroy)
DV_STUB(dv_model_calibrate) DV_STUB(dv_model_apply_corrections) DV_STUB(dv_model_graph_stats) DV_STUB(dv_model_infer) DV_STUB(dv_model_layer_info) DV_STUB(dv_model_load_weights)
DV_STUB(dv_model_num_layers) DV_STUB(dv_model_num_params)
DV_STUB(dv_allele_counts_arrays) DV_STUB(dv_allele_counts_free) DV_STUB(dv_count_alleles) DV_STUB(dv_count_alleles_batch)
DV_STUB(dv_merge_alt_channels)

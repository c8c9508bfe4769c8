//! Raw declarations of include/needle.h and the part of include/needle_hip.h this crate uses.
//! Keep in step with those headers; tests/test_capi_cpu.py checks that the library exports every symbol.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

/// `enum NeedleError` (needle-capi/src/lib.rs:58-85): repr(C), values 0..=11.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum NeedleError {
    Ok = 0,
    InvalidUtf8String,
    NullArgument,
    InvalidArgument,
    FrameHashDataNotFound,
    FrameHashDataInvalidVersion,
    InvalidFrameHashData,
    ComparatorMinimumPaths,
    AnalyzerInvalidHashPeriod,
    AnalyzerInvalidHashDuration,
    IOError,
    Unknown,
}

#[repr(C)]
pub struct NeedleAudioAnalyzer {
    _private: [u8; 0],
}
#[repr(C)]
pub struct NeedleAudioComparator {
    _private: [u8; 0],
}
#[repr(C)]
pub struct FrameHashes {
    _private: [u8; 0],
}

#[repr(C)]
pub struct NeedleHipLibrary {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct NeedleHipSearchResult {
    pub has_result: bool,
    pub has_opening: bool,
    pub has_ending: bool,
    pub opening_start_ns: u64,
    pub opening_end_ns: u64,
    pub ending_start_ns: u64,
    pub ending_end_ns: u64,
}

/// One matched segment of a pair (`NeedleHipRun`, include/needle_hip.h): what
/// `Comparator::longest_common_hash_match` pushes per maximal run (comparator.rs:196-229).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct NeedleHipRun {
    pub problem: u32,
    pub src_end: u32,
    pub dst_end: u32,
    pub len: u32,
    pub src_match_hash: u32,
    pub dst_match_hash: u32,
}

/// Result of `needle_hip_library_audit` (f32 first pass vs the f64 kernel over the same resident PCM).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct NeedleHipCertAudit {
    pub items: u64,
    pub accepted: u64,
    pub accepted_mismatches: u64,
    pub mismatches: u64,
    pub max_error_over_s: f64,
    pub max_s: f64,
}

extern "C" {
    // ---- needle.h ------------------------------------------------------------------------------------
    pub fn needle_error_to_str(error: NeedleError) -> *const c_char;
    pub fn needle_audio_analyzer_new(
        paths: *const *const c_char,
        num_paths: usize,
        opening_search_percentage: f32,
        ending_search_percentage: f32,
        include_endings: bool,
        threaded_decoding: bool,
        force: bool,
        output: *mut *mut NeedleAudioAnalyzer,
    ) -> NeedleError;
    pub fn needle_audio_analyzer_free(analyzer: *const NeedleAudioAnalyzer);
    pub fn needle_audio_analyzer_get_frame_hashes(
        analyzer: *const NeedleAudioAnalyzer,
        index: usize,
        output: *mut *const FrameHashes,
    ) -> NeedleError;
    pub fn needle_audio_analyzer_run(
        analyzer: *mut NeedleAudioAnalyzer,
        hash_duration: f32,
        persist: bool,
        threading: bool,
    ) -> NeedleError;
    pub fn needle_audio_comparator_new(
        paths: *const *const c_char,
        num_paths: usize,
        include_endings: bool,
        hash_match_threshold: u16,
        min_opening_duration: u16,
        min_ending_duration: u16,
        time_padding: f32,
        output: *mut *const NeedleAudioComparator,
    ) -> NeedleError;
    pub fn needle_audio_comparator_free(comparator: *const NeedleAudioComparator);
    pub fn needle_audio_comparator_run(
        comparator: *const NeedleAudioComparator,
        analyze: bool,
        display: bool,
        use_skip_files: bool,
        write_skip_files: bool,
        threading: bool,
    ) -> NeedleError;

    // ---- needle_hip.h --------------------------------------------------------------------------------
    pub fn needle_hip_last_error_message() -> *const c_char;
    pub fn needle_hip_device_count(count: *mut c_int) -> NeedleError;
    pub fn needle_hip_set_device(ordinal: c_int) -> NeedleError;
    pub fn needle_hip_synchronize() -> NeedleError;
    /// PCI address of the current device ("0000:c1:00.0", NUL-terminated) into a 32-byte buffer.
    pub fn needle_hip_device_pci_bus_id(out: *mut c_char) -> NeedleError;
    /// {items, items recomputed in f64, chunks, chunks recomputed} of the certified f32 first pass.
    pub fn needle_hip_fingerprint_cert_stats(counts: *mut u64, reset: bool) -> NeedleError;
    /// Cell evaluations issued by the counting instantiation of the scan (NEEDLE_HIP_SCAN_COUNT=1).
    pub fn needle_hip_scan_issued_evaluations(lane_evaluations: *mut u64, reset: bool) -> NeedleError;
    /// Form of this process's last scan launch (3: aligned windows on the vector ALU, 4: head rows on the matrix pipe)
    /// and, for form 4, the matrix instructions it issued.
    pub fn needle_hip_scan_last_launch(form: *mut i32, matrix_products: *mut u64) -> NeedleError;
    pub fn needle_hip_epilogue_host_fallbacks(jobs: *mut u64, reset: bool) -> NeedleError;
    /// The library's `hipStream_t` on the current device (NULL without one).
    pub fn needle_hip_stream() -> *mut c_void;
    pub fn needle_hip_analyzer_run_pcm(
        analyzer: *mut NeedleAudioAnalyzer,
        pcm: *const *const i16,
        num_values: *const usize,
        channels: c_int,
        sample_rate: c_int,
        hash_duration: f32,
        persist: bool,
    ) -> NeedleError;
    pub fn needle_hip_comparator_run_with_frame_hashes(
        comparator: *const NeedleAudioComparator,
        frame_hashes: *const *const FrameHashes,
        num_videos: usize,
        display: bool,
        use_skip_files: bool,
        write_skip_files: bool,
        results: *mut NeedleHipSearchResult,
    ) -> NeedleError;
    pub fn needle_hip_frame_hashes_new(
        opening_hashes: *const u32,
        opening_ts_ns: *const u64,
        num_opening: usize,
        ending_hashes: *const u32,
        ending_ts_ns: *const u64,
        num_ending: usize,
        hash_duration_ns: u64,
        md5: *const c_char,
        output: *mut *mut FrameHashes,
    ) -> NeedleError;
    pub fn needle_hip_frame_hashes_free(frame_hashes: *mut FrameHashes);
    pub fn needle_hip_frame_hashes_len(frame_hashes: *const FrameHashes, ending: bool) -> usize;
    pub fn needle_hip_frame_hashes_copy(
        frame_hashes: *const FrameHashes,
        ending: bool,
        hashes: *mut u32,
        ts_ns: *mut u64,
        capacity: usize,
    ) -> NeedleError;
    pub fn needle_hip_frame_hashes_hash_duration_ns(frame_hashes: *const FrameHashes) -> u64;
    pub fn needle_hip_frame_hashes_md5(frame_hashes: *const FrameHashes) -> *const c_char;
    pub fn needle_hip_frame_hashes_read(path: *const c_char, output: *mut *mut FrameHashes) -> NeedleError;
    pub fn needle_hip_frame_hashes_write(frame_hashes: *const FrameHashes, path: *const c_char) -> NeedleError;
    // ---- multi-GPU: one process per GPU, RCCL inside the library (include/needle_hip.h "multi-GPU") ----
    pub fn needle_hip_comm_create_id(id: *mut u8) -> NeedleError; // NEEDLE_HIP_COMM_ID_BYTES = 128
    pub fn needle_hip_comm_init(id: *const u8, rank: c_int, world_size: c_int) -> NeedleError;
    pub fn needle_hip_comm_finalize();
    pub fn needle_hip_comm_rank() -> c_int;
    pub fn needle_hip_comm_world_size() -> c_int;
    pub fn needle_hip_comm_barrier() -> NeedleError;
    pub fn needle_hip_comm_all_gather_host(send: *const c_void, recv: *mut c_void, bytes_per_rank: usize) -> NeedleError;
    pub fn needle_hip_comm_shard(units: usize, world_size: c_int, rank: c_int, first: *mut usize, count: *mut usize);
    pub fn needle_hip_library_new(
        num_videos: usize,
        opening_search_percentage: f32,
        hash_duration: f32,
        output: *mut *mut NeedleHipLibrary,
    ) -> NeedleError;
    pub fn needle_hip_library_free(library: *mut NeedleHipLibrary);
    pub fn needle_hip_library_include_endings(library: *mut NeedleHipLibrary, ending_search_percentage: f32) -> NeedleError;
    pub fn needle_hip_library_set_pcm(
        library: *mut NeedleHipLibrary,
        pcm: *const *const i16,
        num_values: *const usize,
        channels: c_int,
    ) -> NeedleError;
    /// Videos `[first_video, first_video + video_count)` whose PCM rank `rank` of `world_size` must hold.
    pub fn needle_hip_library_rank_videos(
        library: *const NeedleHipLibrary,
        num_values: *const usize,
        channels: c_int,
        world_size: c_int,
        rank: c_int,
        first_video: *mut usize,
        video_count: *mut usize,
    ) -> NeedleError;
    /// `d_pcm[i]` are DEVICE pointers (PCM decoded or generated on the GPU), NULL for videos of other ranks.
    pub fn needle_hip_library_set_pcm_device(
        library: *mut NeedleHipLibrary,
        d_pcm: *const *const i16,
        num_values: *const usize,
        channels: c_int,
    ) -> NeedleError;
    pub fn needle_hip_library_stream_pcm(
        library: *mut NeedleHipLibrary,
        pcm: *const *const i16,
        num_values: *const usize,
        channels: c_int,
    ) -> NeedleError;
    pub fn needle_hip_library_job_begin(
        library: *mut NeedleHipLibrary,
        comparator: *const NeedleAudioComparator,
        slot: c_int,
    ) -> NeedleError;
    pub fn needle_hip_library_job_end(
        library: *mut NeedleHipLibrary,
        comparator: *const NeedleAudioComparator,
        slot: c_int,
        results: *mut NeedleHipSearchResult,
        num_runs: *mut usize,
    ) -> NeedleError;
    pub fn needle_hip_library_frame_hashes(
        library: *mut NeedleHipLibrary,
        index: usize,
        output: *mut *mut FrameHashes,
    ) -> NeedleError;
    /// The complete run list of the job that finished last in `slot`; valid until the slot's next `job_begin`.
    pub fn needle_hip_library_job_runs(
        library: *const NeedleHipLibrary,
        slot: c_int,
        runs: *mut *const NeedleHipRun,
        num_runs: *mut usize,
    ) -> NeedleError;
    /// `{hash rows, run heads, results}` bytes received in the job's all-gathers, `[3]` = scans repeated.
    pub fn needle_hip_library_job_comm_bytes(library: *const NeedleHipLibrary, slot: c_int, bytes: *mut u64) -> NeedleError;
    pub fn needle_hip_library_job_form(library: *const NeedleHipLibrary, slot: c_int, form: *mut u32) -> NeedleError;
    pub fn needle_hip_library_audit(library: *mut NeedleHipLibrary, audit: *mut NeedleHipCertAudit) -> NeedleError;
    pub fn needle_hip_host_threads() -> c_int;
    pub fn needle_hip_host_alloc(host_ptr: *mut *mut c_void, bytes: usize) -> NeedleError;
    pub fn needle_hip_host_alloc_free(host_ptr: *mut c_void) -> NeedleError;
}

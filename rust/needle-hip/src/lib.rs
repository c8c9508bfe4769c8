//! The `needle::audio` surface on the MI355X path.
//!
//! `Analyzer`, `Comparator`, `FrameHashes` and `SearchResult` carry the method set of the reference's
//! structs (needle/src/audio/analyzer.rs:95-151,425; comparator.rs:65-147,524,637; data.rs:74-168), so code
//! written against `needle::audio` ports by changing the `use` line.  Two differences, both forced by the
//! boundary this build draws (FFmpeg is upstream of it): video files are RIFF/WAVE PCM, and `Analyzer::run_pcm`
//! exists for callers that decode themselves.
//!
//! UNTESTED — see Cargo.toml.
pub mod ffi;

use std::ffi::{CStr, CString};
use std::path::{Path, PathBuf};
use std::ptr;
use std::time::Duration;

/// needle::Error (needle/src/lib.rs:117-149), as far as it crosses the C ABI (needle-capi/src/lib.rs:121-134).
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Error {
    pub code: ffi::NeedleError,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "{}", self.message)
    }
}
impl std::error::Error for Error {}

pub type Result<T> = std::result::Result<T, Error>;

fn check(code: ffi::NeedleError) -> Result<()> {
    if code == ffi::NeedleError::Ok {
        return Ok(());
    }
    // SAFETY: both functions return pointers to NUL-terminated strings owned by the library.
    let message = unsafe {
        let detail = ffi::needle_hip_last_error_message();
        if !detail.is_null() && *detail != 0 {
            CStr::from_ptr(detail).to_string_lossy().into_owned()
        } else {
            CStr::from_ptr(ffi::needle_error_to_str(code)).to_string_lossy().into_owned()
        }
    };
    Err(Error { code, message })
}

fn c_paths<P: AsRef<Path>>(paths: &[P]) -> Result<(Vec<CString>, Vec<*const std::os::raw::c_char>)> {
    let owned = paths
        .iter()
        .map(|p| {
            CString::new(p.as_ref().to_string_lossy().as_bytes()).map_err(|_| Error {
                code: ffi::NeedleError::InvalidUtf8String,
                message: "path contains a NUL byte".into(),
            })
        })
        .collect::<Result<Vec<_>>>()?;
    let raw = owned.iter().map(|s| s.as_ptr()).collect();
    Ok((owned, raw))
}

pub const DEFAULT_HASH_MATCH_THRESHOLD: u16 = 10; // audio/mod.rs:14
pub const DEFAULT_OPENING_SEARCH_PERCENTAGE: f32 = 0.50;
pub const DEFAULT_ENDING_SEARCH_PERCENTAGE: f32 = 0.25;
pub const DEFAULT_MIN_OPENING_DURATION: u16 = 20;
pub const DEFAULT_MIN_ENDING_DURATION: u16 = 20;
pub const DEFAULT_HASH_DURATION: f32 = 0.3;
pub const DEFAULT_OPENING_AND_ENDING_TIME_PADDING: f32 = 0.0;

/// data.rs:74-80.  Owns a library-side `FrameHashes`.
pub struct FrameHashes {
    raw: *mut ffi::FrameHashes,
    opening: Vec<(u32, Duration)>,
    ending: Vec<(u32, Duration)>,
    md5: String,
}

// The handle is plain heap data inside the library, not tied to a thread.
unsafe impl Send for FrameHashes {}

impl FrameHashes {
    /// Takes ownership of `raw`.
    unsafe fn from_raw(raw: *mut ffi::FrameHashes) -> Result<Self> {
        let side = |ending: bool| -> Result<Vec<(u32, Duration)>> {
            let n = ffi::needle_hip_frame_hashes_len(raw, ending);
            let (mut h, mut t) = (vec![0u32; n], vec![0u64; n]);
            check(ffi::needle_hip_frame_hashes_copy(raw, ending, h.as_mut_ptr(), t.as_mut_ptr(), n))?;
            Ok(h.into_iter().zip(t.into_iter().map(Duration::from_nanos)).collect())
        };
        let md5 = CStr::from_ptr(ffi::needle_hip_frame_hashes_md5(raw)).to_string_lossy().into_owned();
        Ok(FrameHashes { raw, opening: side(false)?, ending: side(true)?, md5 })
    }

    /// Copies a handle the analyzer still owns (needle_audio_analyzer_get_frame_hashes lends it).
    unsafe fn clone_borrowed(borrowed: *const ffi::FrameHashes) -> Result<Self> {
        let side = |ending: bool| -> Result<(Vec<u32>, Vec<u64>)> {
            let n = ffi::needle_hip_frame_hashes_len(borrowed, ending);
            let (mut h, mut t) = (vec![0u32; n], vec![0u64; n]);
            check(ffi::needle_hip_frame_hashes_copy(borrowed, ending, h.as_mut_ptr(), t.as_mut_ptr(), n))?;
            Ok((h, t))
        };
        let (oh, ot) = side(false)?;
        let (eh, et) = side(true)?;
        let mut raw = ptr::null_mut();
        check(ffi::needle_hip_frame_hashes_new(
            oh.as_ptr(),
            ot.as_ptr(),
            oh.len(),
            eh.as_ptr(),
            et.as_ptr(),
            eh.len(),
            ffi::needle_hip_frame_hashes_hash_duration_ns(borrowed),
            ffi::needle_hip_frame_hashes_md5(borrowed),
            &mut raw,
        ))?;
        Self::from_raw(raw)
    }

    /// data.rs:104-115: `<video>.needle.dat` next to the video.
    pub fn from_path(path: impl AsRef<Path>) -> Result<Self> {
        let (_owned, raw_paths) = c_paths(&[path])?;
        let mut raw = ptr::null_mut();
        // SAFETY: valid NUL-terminated path, valid out pointer.
        unsafe {
            check(ffi::needle_hip_frame_hashes_read(raw_paths[0], &mut raw))?;
            Self::from_raw(raw)
        }
    }

    pub fn to_path(&self, path: impl AsRef<Path>) -> Result<()> {
        let (_owned, raw_paths) = c_paths(&[path])?;
        unsafe { check(ffi::needle_hip_frame_hashes_write(self.raw, raw_paths[0])) }
    }

    pub fn opening_data(&self) -> &[(u32, Duration)] {
        &self.opening
    }
    pub fn ending_data(&self) -> &[(u32, Duration)] {
        &self.ending
    }
    pub fn hash_duration(&self) -> Duration {
        Duration::from_nanos(unsafe { ffi::needle_hip_frame_hashes_hash_duration_ns(self.raw) })
    }
    pub fn md5(&self) -> &str {
        &self.md5
    }
}

impl Drop for FrameHashes {
    fn drop(&mut self) {
        unsafe { ffi::needle_hip_frame_hashes_free(self.raw) }
    }
}

/// analyzer.rs:77-151.
pub struct Analyzer<P: AsRef<Path>> {
    videos: Vec<P>,
    opening_search_percentage: f32,
    ending_search_percentage: f32,
    include_endings: bool,
    threaded_decoding: bool,
    force: bool,
}

impl<P: AsRef<Path>> Default for Analyzer<P> {
    fn default() -> Self {
        Analyzer {
            videos: Vec::new(),
            opening_search_percentage: DEFAULT_OPENING_SEARCH_PERCENTAGE,
            ending_search_percentage: DEFAULT_ENDING_SEARCH_PERCENTAGE,
            include_endings: false,
            threaded_decoding: false,
            force: false,
        }
    }
}

impl<P: AsRef<Path>> Analyzer<P> {
    pub fn from_files(videos: impl Into<Vec<P>>, threaded_decoding: bool, force: bool) -> Self {
        Analyzer { videos: videos.into(), threaded_decoding, force, ..Default::default() }
    }
    pub fn videos(&self) -> &[P] {
        &self.videos
    }
    pub fn with_opening_search_percentage(mut self, v: f32) -> Self {
        self.opening_search_percentage = v;
        self
    }
    pub fn with_ending_search_percentage(mut self, v: f32) -> Self {
        self.ending_search_percentage = v;
        self
    }
    pub fn with_include_endings(mut self, v: bool) -> Self {
        self.include_endings = v;
        self
    }
    pub fn with_threaded_decoding(mut self, v: bool) -> Self {
        self.threaded_decoding = v;
        self
    }
    pub fn with_force(mut self, v: bool) -> Self {
        self.force = v;
        self
    }

    fn handle(&self) -> Result<*mut ffi::NeedleAudioAnalyzer> {
        let (_owned, raw) = c_paths(&self.videos)?;
        let mut out = ptr::null_mut();
        // SAFETY: `raw` outlives the call; the library copies the strings (needle-capi/src/lib.rs:373-409).
        unsafe {
            check(ffi::needle_audio_analyzer_new(
                raw.as_ptr(),
                raw.len(),
                self.opening_search_percentage,
                self.ending_search_percentage,
                self.include_endings,
                self.threaded_decoding,
                self.force,
                &mut out,
            ))?;
        }
        Ok(out)
    }

    fn collect(&self, handle: *mut ffi::NeedleAudioAnalyzer) -> Result<Vec<FrameHashes>> {
        (0..self.videos.len())
            .map(|i| unsafe {
                let mut fh = ptr::null();
                check(ffi::needle_audio_analyzer_get_frame_hashes(handle, i, &mut fh))?;
                FrameHashes::clone_borrowed(fh)
            })
            .collect()
    }

    /// analyzer.rs:425-455.  `threading` is accepted for signature parity: the batch is one GPU launch.
    pub fn run(&self, hash_duration: Duration, persist: bool, threading: bool) -> Result<Vec<FrameHashes>> {
        let handle = self.handle()?;
        let result = unsafe {
            check(ffi::needle_audio_analyzer_run(handle, hash_duration.as_secs_f32(), persist, threading))
        }
        .and_then(|_| self.collect(handle));
        unsafe { ffi::needle_audio_analyzer_free(handle) };
        result
    }

    /// Extension: the decoded stream of every video, interleaved s16 at `sample_rate` (any rate; it is
    /// resampled to chromaprint's 11025 Hz on the device), instead of a file to decode.
    pub fn run_pcm(
        &self,
        pcm: &[&[i16]],
        channels: i32,
        sample_rate: i32,
        hash_duration: Duration,
        persist: bool,
    ) -> Result<Vec<FrameHashes>> {
        assert_eq!(pcm.len(), self.videos.len(), "one PCM stream per video");
        let ptrs: Vec<*const i16> = pcm.iter().map(|s| s.as_ptr()).collect();
        let lens: Vec<usize> = pcm.iter().map(|s| s.len()).collect();
        let handle = self.handle()?;
        let result = unsafe {
            check(ffi::needle_hip_analyzer_run_pcm(
                handle,
                ptrs.as_ptr(),
                lens.as_ptr(),
                channels,
                sample_rate,
                hash_duration.as_secs_f32(),
                persist,
            ))
        }
        .and_then(|_| self.collect(handle));
        unsafe { ffi::needle_audio_analyzer_free(handle) };
        result
    }
}

/// comparator.rs:65-69.  The reference keeps the fields private; accessors are an extension.
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct SearchResult {
    opening: Option<(Duration, Duration)>,
    ending: Option<(Duration, Duration)>,
}

impl SearchResult {
    pub fn opening(&self) -> Option<(Duration, Duration)> {
        self.opening
    }
    pub fn ending(&self) -> Option<(Duration, Duration)> {
        self.ending
    }
}

/// comparator.rs:74-147.
pub struct Comparator<P: AsRef<Path>> {
    videos: Vec<P>,
    include_endings: bool,
    hash_match_threshold: u32,
    min_opening_duration: Duration,
    min_ending_duration: Duration,
    time_padding: Duration,
}

impl<P: AsRef<Path>> Default for Comparator<P> {
    fn default() -> Self {
        Comparator {
            videos: Vec::new(),
            include_endings: false,
            hash_match_threshold: DEFAULT_HASH_MATCH_THRESHOLD as u32,
            min_opening_duration: Duration::from_secs(DEFAULT_MIN_OPENING_DURATION as u64),
            min_ending_duration: Duration::from_secs(DEFAULT_MIN_ENDING_DURATION as u64),
            time_padding: Duration::from_secs_f32(DEFAULT_OPENING_AND_ENDING_TIME_PADDING),
        }
    }
}

impl<P: AsRef<Path>> From<Analyzer<P>> for Comparator<P> {
    fn from(analyzer: Analyzer<P>) -> Self {
        Comparator { videos: analyzer.videos, ..Default::default() } // comparator.rs:96-104: the path list only
    }
}

impl<P: AsRef<Path>> Comparator<P> {
    pub fn from_files(videos: impl Into<Vec<P>>) -> Self {
        Comparator { videos: videos.into(), ..Default::default() }
    }
    pub fn videos(&self) -> &[P] {
        &self.videos
    }
    pub fn with_include_endings(mut self, v: bool) -> Self {
        self.include_endings = v;
        self
    }
    pub fn with_hash_match_threshold(mut self, v: u32) -> Self {
        self.hash_match_threshold = v;
        self
    }
    pub fn with_min_opening_duration(mut self, v: Duration) -> Self {
        self.min_opening_duration = v;
        self
    }
    pub fn with_min_ending_duration(mut self, v: Duration) -> Self {
        self.min_ending_duration = v;
        self
    }
    pub fn with_time_padding(mut self, v: Duration) -> Self {
        self.time_padding = v;
        self
    }

    fn handle(&self) -> Result<*const ffi::NeedleAudioComparator> {
        let (_owned, raw) = c_paths(&self.videos)?;
        let mut out = ptr::null();
        // The C constructor takes whole seconds as u16 (needle-capi/src/lib.rs:556-599), like the CLI does.
        unsafe {
            check(ffi::needle_audio_comparator_new(
                raw.as_ptr(),
                raw.len(),
                self.include_endings,
                self.hash_match_threshold.min(u16::MAX as u32) as u16,
                self.min_opening_duration.as_secs().min(u16::MAX as u64) as u16,
                self.min_ending_duration.as_secs().min(u16::MAX as u64) as u16,
                self.time_padding.as_secs_f32(),
                &mut out,
            ))?;
        }
        Ok(out)
    }

    /// comparator.rs:524-629.  One entry per video that matched something, in video order (`:608-617`).
    pub fn run_with_frame_hashes(
        &self,
        frame_hashes: Vec<FrameHashes>,
        display: bool,
        use_skip_files: bool,
        write_skip_files: bool,
        _threading: bool,
    ) -> Result<Vec<SearchResult>> {
        let handle = self.handle()?;
        let raw: Vec<*const ffi::FrameHashes> = frame_hashes.iter().map(|f| f.raw as *const _).collect();
        let mut results = vec![ffi::NeedleHipSearchResult::default(); raw.len()];
        let status = unsafe {
            check(ffi::needle_hip_comparator_run_with_frame_hashes(
                handle,
                raw.as_ptr(),
                raw.len(),
                display,
                use_skip_files,
                write_skip_files,
                results.as_mut_ptr(),
            ))
        };
        unsafe { ffi::needle_audio_comparator_free(handle) };
        status?;
        let span = |a: u64, b: u64| (Duration::from_nanos(a), Duration::from_nanos(b));
        Ok(results
            .into_iter()
            .filter(|r| r.has_result)
            .map(|r| SearchResult {
                opening: r.has_opening.then(|| span(r.opening_start_ns, r.opening_end_ns)),
                ending: r.has_ending.then(|| span(r.ending_start_ns, r.ending_end_ns)),
            })
            .collect())
    }

    /// comparator.rs:637-660: frame hashes from disk (or analyzed in place), then `run_with_frame_hashes`.
    /// Results are observable through stdout / skip files, as upstream (needle-capi/src/lib.rs:634).
    pub fn run(
        &self,
        analyze: bool,
        display: bool,
        use_skip_files: bool,
        write_skip_files: bool,
        threading: bool,
    ) -> Result<()> {
        let handle = self.handle()?;
        let status = unsafe {
            check(ffi::needle_audio_comparator_run(handle, analyze, display, use_skip_files, write_skip_files, threading))
        };
        unsafe { ffi::needle_audio_comparator_free(handle) };
        status
    }
}

/// Number of HIP devices the library sees (0 on a host without a GPU: every compute call then fails loudly).
pub fn device_count() -> Result<i32> {
    let mut n = 0;
    unsafe { check(ffi::needle_hip_device_count(&mut n))? };
    Ok(n)
}

/// Frame-hash file next to a video: `Path::with_extension("needle.dat")` (data.rs:8-13,117-119).
pub fn frame_hash_path(video: impl AsRef<Path>) -> PathBuf {
    video.as_ref().with_extension("needle.dat")
}

/// One process per GPU (include/needle_hip.h, "multi-GPU").  What upstream does with rayon inside one process
/// (analyzer.rs:437-445 over videos, comparator.rs:549-564 over pairs) is spread over the ranks of a communicator
/// inside libneedle_capi.so: RCCL all-gathers on the library's own streams, no host round trips inside a job.
pub mod multi_gpu {
    use super::{check, ffi, Comparator, Result, SearchResult};
    use std::os::raw::c_int;
    use std::time::Duration;

    pub const COMM_ID_BYTES: usize = 128;

    /// Rank 0: the id every other rank needs (hand it over by file, socket, MPI ...).
    pub fn create_id() -> Result<[u8; COMM_ID_BYTES]> {
        let mut id = [0u8; COMM_ID_BYTES];
        unsafe { check(ffi::needle_hip_comm_create_id(id.as_mut_ptr()))? };
        Ok(id)
    }

    /// Collective over all ranks; binds `device` to this process first.
    pub fn init(id: &[u8; COMM_ID_BYTES], rank: usize, world_size: usize, device: usize) -> Result<()> {
        unsafe {
            check(ffi::needle_hip_set_device(device as c_int))?;
            check(ffi::needle_hip_comm_init(id.as_ptr(), rank as c_int, world_size as c_int))
        }
    }

    pub fn finalize() {
        unsafe { ffi::needle_hip_comm_finalize() }
    }

    /// The videos (or pairs) `rank` owns: `[first, first + count)`.
    pub fn shard(units: usize, world_size: usize, rank: usize) -> (usize, usize) {
        let (mut first, mut count) = (0usize, 0usize);
        unsafe { ffi::needle_hip_comm_shard(units, world_size as c_int, rank as c_int, &mut first, &mut count) };
        (first, count)
    }

    /// An analyze + search job over ALL videos of a library, this rank holding the PCM of its own block only.
    pub struct Library {
        raw: *mut ffi::NeedleHipLibrary,
        num_videos: usize,
    }

    impl Library {
        pub fn new(num_videos: usize, opening_search_percentage: f32, hash_duration: Duration) -> Result<Self> {
            let mut raw = std::ptr::null_mut();
            unsafe {
                check(ffi::needle_hip_library_new(num_videos, opening_search_percentage, hash_duration.as_secs_f32(), &mut raw))?
            };
            Ok(Library { raw, num_videos })
        }

        /// The videos `[first, first + count)` whose PCM rank `rank` of `world_size` has to hold: the fingerprinting
        /// is cut by hashes (equal blocks of the arena), not by videos, so no rank idles; a pure function of the
        /// library's parameters and the stream lengths (`needle_hip_library_rank_videos`).
        pub fn rank_videos(&self, num_values: &[usize], channels: usize, world_size: usize, rank: usize) -> Result<(usize, usize)> {
            assert!(num_values.len() == self.num_videos);
            let (mut first, mut count) = (0usize, 0usize);
            unsafe {
                check(ffi::needle_hip_library_rank_videos(
                    self.raw,
                    num_values.as_ptr(),
                    channels as c_int,
                    world_size as c_int,
                    rank as c_int,
                    &mut first,
                    &mut count,
                ))?
            };
            Ok((first, count))
        }

        /// `pcm[v]` is `None` for the videos another rank owns; `num_values[v]` is known to every rank.
        /// `resident`: keep the PCM in HBM (repeatable analyze) instead of streaming it through.
        pub fn load_pcm(&mut self, pcm: &[Option<&[i16]>], num_values: &[usize], channels: usize, resident: bool) -> Result<()> {
            assert!(pcm.len() == self.num_videos && num_values.len() == self.num_videos);
            let ptrs: Vec<*const i16> = pcm.iter().map(|p| p.map_or(std::ptr::null(), |s| s.as_ptr())).collect();
            unsafe {
                if resident {
                    check(ffi::needle_hip_library_set_pcm(self.raw, ptrs.as_ptr(), num_values.as_ptr(), channels as c_int))
                } else {
                    check(ffi::needle_hip_library_stream_pcm(self.raw, ptrs.as_ptr(), num_values.as_ptr(), channels as c_int))
                }
            }
        }

        /// `Comparator::run_with_frame_hashes` across the communicator: the same `Vec<Option<SearchResult>>` on every rank.
        pub fn run<P: AsRef<std::path::Path>>(&mut self, comparator: &Comparator<P>) -> Result<Vec<Option<SearchResult>>> {
            let handle = comparator.handle()?;
            let mut results = vec![ffi::NeedleHipSearchResult::default(); self.num_videos];
            let status = unsafe {
                check(ffi::needle_hip_library_job_begin(self.raw, handle, 0)).and_then(|_| {
                    check(ffi::needle_hip_library_job_end(self.raw, handle, 0, results.as_mut_ptr(), std::ptr::null_mut()))
                })
            };
            unsafe { ffi::needle_audio_comparator_free(handle) };
            status?;
            let span = |a: u64, b: u64| (Duration::from_nanos(a), Duration::from_nanos(b));
            Ok(results
                .into_iter()
                .map(|r| {
                    r.has_result.then(|| SearchResult {
                        opening: r.has_opening.then(|| span(r.opening_start_ns, r.opening_end_ns)),
                        ending: r.has_ending.then(|| span(r.ending_start_ns, r.ending_end_ns)),
                    })
                })
                .collect())
        }
    }

    impl Library {
        /// The matched segments of every pair -- all ranks' shares -- that the last `run` worked from (a copy).
        pub fn last_runs(&self) -> Result<Vec<ffi::NeedleHipRun>> {
            let (mut ptr, mut n) = (std::ptr::null(), 0usize);
            unsafe { check(ffi::needle_hip_library_job_runs(self.raw, 0, &mut ptr, &mut n))? };
            Ok(if n == 0 { Vec::new() } else { unsafe { std::slice::from_raw_parts(ptr, n) }.to_vec() })
        }

        /// Both transforms (f32 first pass, f64 kernel) over this rank's resident PCM, every kept item compared on the
        /// device: `mismatches` and `accepted_mismatches` must be 0 (`needle_hip_library_audit`).
        pub fn audit(&mut self) -> Result<ffi::NeedleHipCertAudit> {
            let mut a = ffi::NeedleHipCertAudit::default();
            unsafe { check(ffi::needle_hip_library_audit(self.raw, &mut a))? };
            Ok(a)
        }
    }

    impl Drop for Library {
        fn drop(&mut self) {
            unsafe { ffi::needle_hip_library_free(self.raw) }
        }
    }
}

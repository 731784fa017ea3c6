//! Rust binding of the C ABI in `include/audiomatch.h` and the adapter that makes it a
//! third implementor of the reference's `CorrelateAlgo<f32>` (src/matcher/audio_matcher.rs:65-76)
//! next to `LibConvolve` and `MyConvolve`, plus the fast replacement of `calc_chunks`
//! (src/matcher/audio_matcher.rs:88-141).
//!
//! Source only: the build image has no Rust toolchain.

use std::os::raw::{c_char, c_int};
use std::time::Duration;

#[repr(C)]
pub struct AmNeedle {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct AmPeak {
    pub start: u64,
    pub end: u64,
    pub height: f32,
    pub prominence: f32,
}

#[repr(C)]
pub struct AmMatchParams {
    pub sr: u32,
    pub chunk: u64,
    pub overlap: u64,
    pub min_prominence: f32,
    pub min_distance: u64,
    pub overshadow_distance_s: f64,
    pub scale: c_int,
}

pub const AM_OK: c_int = 0;
pub const AM_ERR_CAPACITY: c_int = 2;

extern "C" {
    pub fn am_last_error_string() -> *const c_char;
    pub fn am_needle_create(device: c_int, needle: *const f32, n: usize, out: *mut *mut AmNeedle) -> c_int;
    pub fn am_needle_destroy(h: *mut AmNeedle);
    pub fn am_needle_inv_autocorr(h: *const AmNeedle, out: *mut f32) -> c_int;
    pub fn am_correlate_len(w: usize, s: usize, mode: c_int, out_len: *mut usize) -> c_int;
    pub fn am_correlate(
        h: *const AmNeedle, within: *const f32, w: usize, mode: c_int, scale: c_int,
        out: *mut f32, cap: usize, out_len: *mut usize,
    ) -> c_int;
    pub fn am_match(
        h: *const AmNeedle, haystack: *const f32, len: usize, p: *const AmMatchParams,
        out: *mut AmPeak, cap: usize, n_out: *mut usize,
    ) -> c_int;
    /// interleaved i16 stereo frames, i.e. what minimp3 yields before the down-mix of
    /// mp3_reader.rs:28-37 (the down-mix then happens inside the first kernel, bit-exact)
    pub fn am_needle_create_pcm16(device: c_int, interleaved: *const i16, frames: usize, out: *mut *mut AmNeedle) -> c_int;
    pub fn am_match_pcm16(
        h: *const AmNeedle, interleaved: *const i16, frames: usize, p: *const AmMatchParams,
        out: *mut AmPeak, cap: usize, n_out: *mut usize,
    ) -> c_int;
    /// haystack k -> shard k mod n_shards (no device needed)
    pub fn am_shard_plan(n_items: usize, n_shards: usize, shard: usize, first: *mut usize, stride: *mut usize, count: *mut usize) -> c_int;
    /// the file loop of matcher::run (matcher/mod.rs:42-87) over every GPU of the node
    pub fn am_pool_create(needle: *const f32, n: usize, devices: *const c_int, n_dev: usize, out: *mut *mut AmPool) -> c_int;
    pub fn am_pool_destroy(pool: *mut AmPool);
    pub fn am_pool_match_batch(
        pool: *mut AmPool, haystacks: *const *const f32, lens: *const usize, n_hay: usize,
        p: *const AmMatchParams, out: *mut AmPeak, cap_per_hay: usize, n_out: *mut usize,
    ) -> c_int;
    /// the same loop on the decoder's interleaved i16 stereo frames (mp3_reader.rs:26-37)
    pub fn am_pool_match_batch_pcm16(
        pool: *mut AmPool, interleaved: *const *const i16, frames: *const usize, n_hay: usize,
        p: *const AmMatchParams, out: *mut AmPeak, cap_per_hay: usize, n_out: *mut usize,
    ) -> c_int;
    /// ONE long haystack over the pool's devices (the window fan-out of audio_matcher.rs:104-131 across GPUs,
    /// one sort + overshadow pass over the union, :132-140); sample_format 0 = f32 mono, 1 = i16 stereo frames
    pub fn am_pool_match_long(
        pool: *mut AmPool, haystack: *const std::ffi::c_void, len: usize, sample_format: c_int,
        p: *const AmMatchParams, out: *mut AmPeak, cap: usize, n_out: *mut usize,
    ) -> c_int;
    /// the pieces of that, for one process per GPU: the split as a pure function, one part (peaks unmerged), the merge
    pub fn am_long_plan(
        len: usize, needle_len: usize, p: *const AmMatchParams, n_parts: usize, part: usize,
        first_window: *mut usize, n_windows: *mut usize, first_sample: *mut usize, n_samples: *mut usize,
    ) -> c_int;
    pub fn am_match_part_device(
        h: *const AmNeedle, d_part: *const std::ffi::c_void, n_samples: usize, sample_format: c_int, p: *const AmMatchParams,
        n_windows: usize, first_sample: u64, out: *mut AmPeak, cap: usize, n_out: *mut usize,
    ) -> c_int;
    pub fn am_merge_peaks(p: *const AmMatchParams, peaks: *const AmPeak, n: usize, out: *mut AmPeak, cap: usize, n_out: *mut usize) -> c_int;
    /// pinned host memory for the decoder's output (read by the copy engines without a bounce buffer)
    pub fn am_host_alloc(bytes: usize, out: *mut *mut std::ffi::c_void) -> c_int;
    pub fn am_host_free(p: *mut std::ffi::c_void) -> c_int;
    pub fn am_host_register(p: *mut std::ffi::c_void, bytes: usize) -> c_int;
    pub fn am_host_unregister(p: *mut std::ffi::c_void) -> c_int;
    /// several snippets of one length: the haystack's forward transform is shared by a group of needles
    pub fn am_match_multi_batch_device(
        needles: *const *const AmNeedle, n_needles: usize, d_haystacks: *const *const std::ffi::c_void,
        lens: *const usize, n_hay: usize, sample_format: c_int, p: *const AmMatchParams,
        out: *mut AmPeak, cap_per_pair: usize, n_out: *mut usize,
    ) -> c_int;
    pub fn am_pool_create_multi(
        needles: *const *const f32, n_needles: usize, n: usize, devices: *const c_int, n_dev: usize, out: *mut *mut AmPool,
    ) -> c_int;
    pub fn am_pool_match_multi_batch(
        pool: *mut AmPool, haystacks: *const *const std::ffi::c_void, lens: *const usize, n_hay: usize, sample_format: c_int,
        p: *const AmMatchParams, out: *mut AmPeak, cap_per_pair: usize, n_out: *mut usize,
    ) -> c_int;
    /// calc_chunks on the lazy sample iterator (audio_matcher.rs:88-104, mp3_reader.rs:13-41)
    pub fn am_match_stream_begin(
        h: *const AmNeedle, sample_format: c_int, expected_len: usize, p: *const AmMatchParams, out: *mut *mut AmStream,
    ) -> c_int;
    pub fn am_match_stream_push(st: *mut AmStream, samples: *const std::ffi::c_void, n: usize) -> c_int;
    pub fn am_match_stream_finish(st: *mut AmStream, out: *mut AmPeak, cap: usize, n_out: *mut usize) -> c_int;
    pub fn am_match_stream_destroy(st: *mut AmStream);
}

pub const AM_FMT_F32_MONO: c_int = 0;
pub const AM_FMT_S16_STEREO: c_int = 1;

#[repr(C)]
pub struct AmStream {
    _private: [u8; 0],
}

#[repr(C)]
pub struct AmPool {
    _private: [u8; 0],
}

/// audio_matcher.rs:55-59
#[derive(Debug, Clone, Copy)]
pub enum Mode {
    Full,
    Same,
    Valid,
}

fn am_err(rc: c_int) -> Box<dyn std::error::Error> {
    let msg = unsafe { std::ffi::CStr::from_ptr(am_last_error_string()) }.to_string_lossy().into_owned();
    format!("audiomatch error {rc}: {msg}").into()
}

/// Drop-in for `LibConvolve` (matcher/mod.rs:34).  In the reference crate this type gets
/// `impl CorrelateAlgo<SampleType> for HipConvolve` with exactly these two methods.
pub struct HipConvolve {
    h: *mut AmNeedle,
    len: usize,
}
unsafe impl Send for HipConvolve {}
unsafe impl Sync for HipConvolve {}

impl HipConvolve {
    pub fn new(sample_data: Box<[f32]>) -> Result<Self, Box<dyn std::error::Error>> {
        let mut h = std::ptr::null_mut();
        let rc = unsafe { am_needle_create(0, sample_data.as_ptr(), sample_data.len(), &mut h) };
        if rc != AM_OK {
            return Err(am_err(rc));
        }
        Ok(Self { h, len: sample_data.len() })
    }

    /// The same from decoded stereo PCM (`frame.data`, mp3_reader.rs:28): no CPU down-mix pass.
    pub fn from_pcm16(interleaved: &[i16]) -> Result<Self, Box<dyn std::error::Error>> {
        let mut h = std::ptr::null_mut();
        let frames = interleaved.len() / 2;
        let rc = unsafe { am_needle_create_pcm16(0, interleaved.as_ptr(), frames, &mut h) };
        if rc != AM_OK {
            return Err(am_err(rc));
        }
        Ok(Self { h, len: frames })
    }

    /// CorrelateAlgo::inverse_sample_auto_correlation (audio_matcher.rs:66)
    pub fn inverse_sample_auto_correlation(&self) -> f32 {
        let mut v = 0f32;
        unsafe { am_needle_inv_autocorr(self.h, &mut v) };
        v
    }

    /// CorrelateAlgo::correlate_with_sample (audio_matcher.rs:67-72)
    pub fn correlate_with_sample(&self, within: &[f32], mode: Mode, scale: bool) -> Result<Vec<f32>, Box<dyn std::error::Error>> {
        let m = match mode {
            Mode::Full => 0,
            Mode::Same => 1,
            Mode::Valid => 2,
        };
        let mut n = 0usize;
        let rc = unsafe { am_correlate_len(within.len(), self.len, m, &mut n) };
        if rc != AM_OK {
            return Err(am_err(rc));
        }
        let mut out = vec![0f32; n];
        let rc = unsafe { am_correlate(self.h, within.as_ptr(), within.len(), m, scale as c_int, out.as_mut_ptr(), out.len(), &mut n) };
        if rc != AM_OK {
            return Err(am_err(rc));
        }
        Ok(out)
    }

    /// calc_chunks (audio_matcher.rs:88-141) on the GPU: peaks sorted by start, overshadowed ones removed.
    pub fn calc_chunks(&self, sr: u16, m_samples: &[f32], scale: bool, chunk_size: Duration, overlap_length: Duration,
                       distance: Duration, prominence: f32) -> Result<Vec<AmPeak>, Box<dyn std::error::Error>> {
        let p = AmMatchParams {
            sr: sr as u32,
            chunk: (chunk_size.as_secs_f64() * sr as f64).round() as u64,       // :100
            overlap: (overlap_length.as_secs_f64() * sr as f64).round() as u64, // :99
            min_prominence: prominence,                                         // :227
            min_distance: distance.as_secs() * sr as u64,                       // :228
            overshadow_distance_s: distance.as_secs_f64(),                      // :137-138
            scale: scale as c_int,
        };
        let mut buf = vec![AmPeak::default(); 256];
        let mut n = 0usize;
        let mut rc = unsafe { am_match(self.h, m_samples.as_ptr(), m_samples.len(), &p, buf.as_mut_ptr(), buf.len(), &mut n) };
        if rc == AM_ERR_CAPACITY {
            buf.resize(n, AmPeak::default());
            rc = unsafe { am_match(self.h, m_samples.as_ptr(), m_samples.len(), &p, buf.as_mut_ptr(), buf.len(), &mut n) };
        }
        if rc != AM_OK {
            return Err(am_err(rc));
        }
        buf.truncate(n);
        Ok(buf)
    }

    /// `calc_chunks` on decoded stereo PCM frames (interleaved i16, as minimp3 delivers them):
    /// same result as down-mixing on the CPU first, one pass less over the haystack.
    pub fn calc_chunks_pcm16(&self, p: &AmMatchParams, interleaved: &[i16]) -> Result<Vec<AmPeak>, Box<dyn std::error::Error>> {
        let frames = interleaved.len() / 2;
        let mut buf = vec![AmPeak::default(); 256];
        let mut n = 0usize;
        let mut rc = unsafe { am_match_pcm16(self.h, interleaved.as_ptr(), frames, p, buf.as_mut_ptr(), buf.len(), &mut n) };
        if rc == AM_ERR_CAPACITY {
            buf.resize(n, AmPeak::default());
            rc = unsafe { am_match_pcm16(self.h, interleaved.as_ptr(), frames, p, buf.as_mut_ptr(), buf.len(), &mut n) };
        }
        if rc != AM_OK {
            return Err(am_err(rc));
        }
        buf.truncate(n);
        Ok(buf)
    }
}

impl Drop for HipConvolve {
    fn drop(&mut self) {
        unsafe { am_needle_destroy(self.h) }
    }
}

/// The file loop of `matcher::run` (matcher/mod.rs:42-87) over every GPU of the node: the needle
/// replicated per device, haystack `k` matched on device `k mod n` (no exchange step), one submit
/// thread per device inside the library, every result in the slot of its haystack.
pub struct HipConvolvePool {
    p: *mut AmPool,
}
unsafe impl Send for HipConvolvePool {}

impl HipConvolvePool {
    /// `devices = None`: every visible device
    pub fn new(sample_data: &[f32], devices: Option<&[c_int]>) -> Result<Self, Box<dyn std::error::Error>> {
        let mut p = std::ptr::null_mut();
        let (dp, dn) = match devices {
            Some(d) => (d.as_ptr(), d.len()),
            None => (std::ptr::null(), 0),
        };
        let rc = unsafe { am_pool_create(sample_data.as_ptr(), sample_data.len(), dp, dn, &mut p) };
        if rc != AM_OK {
            return Err(am_err(rc));
        }
        Ok(Self { p })
    }

    /// `calc_chunks` for every haystack of the batch; result `k` belongs to `haystacks[k]`
    pub fn calc_chunks(&mut self, p: &AmMatchParams, haystacks: &[&[f32]]) -> Result<Vec<Vec<AmPeak>>, Box<dyn std::error::Error>> {
        let ptrs: Vec<*const f32> = haystacks.iter().map(|h| h.as_ptr()).collect();
        let lens: Vec<usize> = haystacks.iter().map(|h| h.len()).collect();
        let mut cap = 64usize;
        loop {
            let mut buf = vec![AmPeak::default(); cap * haystacks.len().max(1)];
            let mut n = vec![0usize; haystacks.len()];
            let rc = unsafe {
                am_pool_match_batch(self.p, ptrs.as_ptr(), lens.as_ptr(), haystacks.len(), p, buf.as_mut_ptr(), cap, n.as_mut_ptr())
            };
            if rc == AM_ERR_CAPACITY {
                cap = n.iter().copied().max().unwrap_or(cap).max(cap + 1);
                continue;
            }
            if rc != AM_OK {
                return Err(am_err(rc));
            }
            return Ok((0..haystacks.len()).map(|k| buf[k * cap..k * cap + n[k]].to_vec()).collect());
        }
    }

    /// calc_chunks on ONE long recording, its windows split over the pool's devices
    pub fn match_long(&self, haystack: &[f32], p: &AmMatchParams) -> Result<Vec<AmPeak>, Box<dyn std::error::Error>> {
        let mut cap = 4096usize;
        loop {
            let mut buf = vec![AmPeak::default(); cap];
            let mut n = 0usize;
            let rc = unsafe { am_pool_match_long(self.p, haystack.as_ptr().cast(), haystack.len(), 0, p, buf.as_mut_ptr(), cap, &mut n) };
            if rc == AM_ERR_CAPACITY {
                cap = n;
                continue;
            }
            if rc != AM_OK {
                return Err(am_err(rc));
            }
            buf.truncate(n);
            return Ok(buf);
        }
    }
}

impl Drop for HipConvolvePool {
    fn drop(&mut self) {
        unsafe { am_pool_destroy(self.p) }
    }
}

impl HipConvolve {
    /// `calc_chunks` on the reference's own argument shape: a lazy `ExactSizeIterator` of samples
    /// (audio_matcher.rs:88-97).  Blocks of samples are pushed as the iterator yields them; the
    /// transforms of every block pair that has arrived run while the decoder is still producing.
    pub fn calc_chunks_iter<I: ExactSizeIterator<Item = f32>>(&self, p: &AmMatchParams, m_samples: I)
        -> Result<Vec<AmPeak>, Box<dyn std::error::Error>> {
        let mut st = std::ptr::null_mut();
        let rc = unsafe { am_match_stream_begin(self.h, AM_FMT_F32_MONO, m_samples.len(), p, &mut st) };
        if rc != AM_OK { return Err(am_err(rc)); }
        let mut block: Vec<f32> = Vec::with_capacity(1 << 20);
        let mut push = |b: &mut Vec<f32>| -> c_int {
            let rc = unsafe { am_match_stream_push(st, b.as_ptr().cast(), b.len()) };
            b.clear();
            rc
        };
        let mut rc = AM_OK;
        for x in m_samples {
            block.push(x);
            if block.len() == block.capacity() { rc = push(&mut block); if rc != AM_OK { break; } }
        }
        if rc == AM_OK { rc = push(&mut block); }
        let mut buf = vec![AmPeak::default(); 256];
        let mut n = 0usize;
        if rc == AM_OK { rc = unsafe { am_match_stream_finish(st, buf.as_mut_ptr(), buf.len(), &mut n) }; }
        unsafe { am_match_stream_destroy(st) };
        if rc != AM_OK { return Err(am_err(rc)); }
        buf.truncate(n);
        Ok(buf)
    }
}

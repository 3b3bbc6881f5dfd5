//! stepth_gpu.rs -- Rust binding of the MI355X engine's C-ABI (`include/stepth_amd.h`, ABI 4,
//! `libstepth_amd.so`) and the bodies a maintainer of nikilark/stepth would put behind the
//! crate's unchanged public API.
//!
//! Replaces: the body of `DepthImage::load_depth_from_additional`
//! (/root/reference/src/depth_image.rs:91-136; signature :91-95 kept) and, for the README's second
//! half, `invert_depth` / `depth_split` / `slice` / `select_foreground`
//! (/root/reference/src/depth_image.rs:162-247) and `MaskImage::apply_mask`
//! (/root/reference/src/mask_image.rs:205-213).  `open_depth_from_additional`
//! (/root/reference/src/depth_image.rs:76-89) keeps decoding the file and delegates as before.
//!
//! SOURCE ONLY: the build image has no rustc/cargo, so this file has never been compiled.  What
//! keeps it honest is `tests/test_rust_binding.py`: it parses the `extern "C"` block and the
//! `#[repr(C)]` structs below and compares names, argument counts and types, field order and
//! sizes, and the constants with `include/stepth_amd.h` -- the header cannot move without this
//! file following.  The compiled and tested counterparts over the same C-ABI are the C++ mirror
//! (`include/stepth_depth_image.hpp`) and the Python mirror (`stepth_amd/depth_image.py`).
//!
//! build.rs of the crate:  println!("cargo:rustc-link-lib=dylib=stepth_amd");
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const STEPTH_ABI_VERSION: u32 = 4;

// stepth_status
pub const STEPTH_OK: c_int = 0;
pub const STEPTH_ERR_BAD_ARG: c_int = -1;
pub const STEPTH_ERR_SIZE_MISMATCH: c_int = -2;
pub const STEPTH_ERR_HIP: c_int = -3;
pub const STEPTH_ERR_MAX_ZERO: c_int = -4;
pub const STEPTH_ERR_NO_DEVICE: c_int = -5;
pub const STEPTH_ERR_OUT_OF_MEMORY: c_int = -6;
pub const STEPTH_ERR_SPLIT_STEP: c_int = -7;
pub const STEPTH_ERR_SPLIT_EMPTY_ZONE: c_int = -8;
pub const STEPTH_ERR_SPLIT_DIVERGED: c_int = -9;
// stepth_mode, stepth_memory, flags
pub const STEPTH_MODE_RING_FIRST_HIT: u32 = 0;
pub const STEPTH_MODE_SAD_WINDOW: u32 = 1;
pub const STEPTH_MEM_HOST: u32 = 0;
pub const STEPTH_MEM_DEVICE: u32 = 1;
pub const STEPTH_FLAG_ASYNC: u32 = 1 << 0;
pub const STEPTH_FLAG_RAW_ONLY: u32 = 1 << 1;
pub const STEPTH_FLAG_GAUSSIAN_TAIL: u32 = 1 << 2;
pub const STEPTH_FLAG_HEAVY_FIRST: u32 = 1 << 3;

/// opaque engine handle (one device, one stream; re-entrant per handle)
#[repr(C)]
pub struct stepth_engine {
    _private: [u8; 0],
}

/// disage::DiscretePixel{position, size, value} as consumed at src/depth_image.rs:110-121
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct stepth_group {
    pub x: u32,
    pub y: u32,
    pub w: u32,
    pub h: u32,
    pub value: [u8; 3],
    pub pad: u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct stepth_image {
    pub data: *const c_void,
    pub width: u32,
    pub height: u32,
    pub channels: u32,
    pub memory: u32,
    pub stride: usize,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct stepth_params {
    pub struct_size: u32,
    pub mode: u32,
    pub precision: [u8; 3],
    pub reserved0: u8,
    pub max_steps: u32,
    pub tile_w: u32,
    pub tile_h: u32,
    pub d_max: u32,
    pub win: u32,
    pub row_begin: u32,
    pub row_end: u32,
    pub flags: u32,
    pub row_block: u32,
    pub row_mod: u32,
    pub row_rem: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct stepth_stats {
    pub groups: u64,
    pub ring0_hits: u64,
    pub probes: u64,
    pub near_steps: u64,
    pub survey_steps: u64,
    pub superblock_steps: u64,
    pub blocks_probed: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct stepth_zone {
    pub has_from: u8,
    pub from: u8,
    pub has_to: u8,
    pub to: u8,
}

extern "C" {
    pub fn stepth_abi_version() -> u32;
    pub fn stepth_status_string(status: c_int) -> *const c_char;
    pub fn stepth_engine_create(device_id: c_int, out: *mut *mut stepth_engine) -> c_int;
    pub fn stepth_engine_destroy(engine: *mut stepth_engine);
    pub fn stepth_engine_set_stream(engine: *mut stepth_engine, hip_stream: *mut c_void) -> c_int;
    pub fn stepth_engine_reset_stream(engine: *mut stepth_engine) -> c_int;
    pub fn stepth_engine_sync(engine: *mut stepth_engine) -> c_int;
    pub fn stepth_last_error(engine: *const stepth_engine) -> *const c_char;
    pub fn stepth_depth_from_additional(engine: *mut stepth_engine, main: *const stepth_image,
        additional: *const stepth_image, params: *const stepth_params, groups: *const stepth_group,
        n_groups: usize, groups_memory: u32, depth: *const stepth_image, max_out: *mut u32) -> c_int;
    pub fn stepth_adaptive_groups(engine: *mut stepth_engine, main: *const stepth_image,
        precision: *const u8, min_splits: u32, max_splits: u32, groups: *mut stepth_group,
        capacity: usize, groups_memory: u32, n_out: *mut usize) -> c_int;
    pub fn stepth_depth_from_additional_batch(engine: *mut stepth_engine, n_frames: u32,
        mains: *const stepth_image, additionals: *const stepth_image, params: *const stepth_params,
        depths: *const stepth_image, max_out: *mut u32) -> c_int;
    pub fn stepth_adaptive_depth_batch(engine: *mut stepth_engine, n_frames: u32,
        mains: *const stepth_image, additionals: *const stepth_image, params: *const stepth_params,
        min_splits: u32, max_splits: u32, depths: *const stepth_image, max_out: *mut u32,
        n_groups_out: *mut usize) -> c_int;
    pub fn stepth_adaptive_groups_batch(engine: *mut stepth_engine, n_frames: u32,
        mains: *const stepth_image, precision: *const u8, min_splits: u32, max_splits: u32,
        groups_out: *const *mut stepth_group, capacity: usize, n_out: *mut usize) -> c_int;
    pub fn stepth_depth_from_groups_batch(engine: *mut stepth_engine, n_frames: u32,
        mains: *const stepth_image, additionals: *const stepth_image, params: *const stepth_params,
        groups: *const *const stepth_group, n_groups: *const usize, depths: *const stepth_image,
        max_out: *mut u32) -> c_int;
    pub fn stepth_host_alloc(out: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn stepth_host_free(ptr: *mut c_void);
    pub fn stepth_normalise_rows(engine: *mut stepth_engine, rows: *const stepth_image,
        max_ptr: *const u32, max_memory: u32, flags: u32) -> c_int;
    pub fn stepth_normalise_rows_pattern(engine: *mut stepth_engine, rows: *const stepth_image,
        first_row: u32, row_block: u32, row_mod: u32, row_rem: u32, max_ptr: *const u32,
        max_memory: u32, flags: u32) -> c_int;
    pub fn stepth_normalise_rows_batch(engine: *mut stepth_engine, n_frames: u32,
        rows: *const stepth_image, first_row: u32, row_block: u32, row_mod: u32, row_rem: u32,
        max_ptr: *const u32, flags: u32) -> c_int;
    pub fn stepth_gaussian_same_size(engine: *mut stepth_engine, src: *const stepth_image,
        dst: *const stepth_image, flags: u32) -> c_int;
    pub fn stepth_distance_dot_array_batch(engine: *mut stepth_engine, array: *const stepth_image,
        what: *const u8, from_xy: *const u32, n: usize, max: u32, precision: *const u8,
        found: *mut u8, dist: *mut u32, pos_xy: *mut u32) -> c_int;
    pub fn stepth_invert_depth(engine: *mut stepth_engine, depth: *const stepth_image, flags: u32) -> c_int;
    pub fn stepth_depth_split(engine: *mut stepth_engine, depth: *const stepth_image, zones: u32,
        out: *mut stepth_zone, capacity: u32, n_out: *mut u32) -> c_int;
    pub fn stepth_slice(engine: *mut stepth_engine, depth: *const stepth_image,
        range: *const stepth_zone, mask: *const stepth_image, flags: u32) -> c_int;
    pub fn stepth_select_foreground(engine: *mut stepth_engine, depth: *const stepth_image,
        mask: *const stepth_image, zone_out: *mut stepth_zone) -> c_int;
    pub fn stepth_select_foreground_batch(engine: *mut stepth_engine, n_frames: u32,
        depths: *const stepth_image, masks: *const stepth_image, zones_out: *mut stepth_zone,
        statuses: *mut c_int, flags: u32) -> c_int;
    pub fn stepth_depth_histogram(engine: *mut stepth_engine, depth: *const stepth_image,
        counts: *mut u32, counts_memory: u32, flags: u32) -> c_int;
    pub fn stepth_zones_from_histogram(engine: *mut stepth_engine, counts: *const u32,
        counts_memory: u32, zones: u32, out: *mut stepth_zone, capacity: u32, n_out: *mut u32) -> c_int;
    pub fn stepth_apply_mask(engine: *mut stepth_engine, rgba: *const stepth_image,
        mask: *const stepth_image, flags: u32) -> c_int;
    pub fn stepth_stats_enable(engine: *mut stepth_engine, enable: c_int) -> c_int;
    pub fn stepth_get_stats(engine: *mut stepth_engine, out: *mut stepth_stats) -> c_int;
    pub fn stepth_get_stats_raw(engine: *mut stepth_engine, out: *mut u64, n: u32) -> c_int;
    pub fn stepth_kernel_time_enable(engine: *mut stepth_engine, enable: c_int) -> c_int;
    pub fn stepth_kernel_time_read(engine: *mut stepth_engine, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn stepth_selftest_exception_guard(engine: *mut stepth_engine, kind: c_int) -> c_int;
}

// ---------------------------------------------------------------------------------------------
// Safe wrapper: one engine per thread that uses it (the C library is re-entrant per handle; the
// reference's `&mut self` makes a call exclusive on its image, src/depth_image.rs:91).
// ---------------------------------------------------------------------------------------------
pub struct Engine(*mut stepth_engine);

unsafe impl Send for Engine {}

impl Engine {
    pub fn new(device_id: i32) -> Result<Engine, String> {
        assert_eq!(unsafe { stepth_abi_version() }, STEPTH_ABI_VERSION, "libstepth_amd.so ABI");
        let mut p = std::ptr::null_mut();
        let rc = unsafe { stepth_engine_create(device_id, &mut p) };
        if rc != STEPTH_OK {
            let s = unsafe { std::ffi::CStr::from_ptr(stepth_status_string(rc)) };
            return Err(s.to_string_lossy().into_owned());
        }
        Ok(Engine(p))
    }
    pub fn raw(&self) -> *mut stepth_engine {
        self.0
    }
    pub fn last_error(&self) -> String {
        unsafe { std::ffi::CStr::from_ptr(stepth_last_error(self.0)) }.to_string_lossy().into_owned()
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { stepth_engine_destroy(self.0) }
    }
}

thread_local! {
    /// the engine the methods below use (created on first use; device 0)
    pub static ENGINE: Engine = Engine::new(0).expect("no MI355X engine: the hot path has no CPU fallback");
}

fn io_invalid(msg: &str) -> std::io::Error {
    std::io::Error::new(std::io::ErrorKind::InvalidInput, msg)
}

fn host_image(data: *const u8, width: u32, height: u32, channels: u32) -> stepth_image {
    stepth_image { data: data as *const c_void, width, height, channels, memory: STEPTH_MEM_HOST,
                   stride: (width * channels) as usize }
}

// ---------------------------------------------------------------------------------------------
// What goes into src/depth_image.rs (the crate's types: `DepthImage { image: RgbaImage, depth:
// GrayImage }`, `MaskImage`; `disage`, `image` as in the crate).  Kept as a comment block so that
// this file stands alone; the signatures are the reference's, src/depth_image.rs:76-80, :91-95.
// ---------------------------------------------------------------------------------------------
/*
impl DepthImage {
    pub fn load_depth_from_additional(&mut self, add_image: image::DynamicImage, precision: [u8; 3])
        -> Result<(), std::io::Error>
    {
        let add = add_image.to_rgb8();                                           // :96
        // disage's grouping stays on the CPU side as it is (:101-110) -- or is replaced by
        // stepth_adaptive_groups(), the engine's own stand-in (parity with disage unpinned)
        let discr = disage::open::rgb_discrete(/* as at :104-109 */);
        let groups: Vec<stepth_group> = discr.pixels().iter().map(|p| stepth_group {
            x: p.position.x, y: p.position.y, w: p.size.width, h: p.size.height,
            value: *p.value, pad: 0 }).collect();                                // :110-121
        let (w, h) = (self.width(), self.height());
        let mut depth = vec![0u8; (w * h) as usize];
        let main = host_image(self.image.as_raw().as_ptr(), w, h, 4);
        let addi = host_image(add.as_raw().as_ptr(), add.width(), add.height(), 3);
        let out = host_image(depth.as_mut_ptr(), w, h, 1);
        let params = stepth_params {
            struct_size: std::mem::size_of::<stepth_params>() as u32,
            mode: STEPTH_MODE_RING_FIRST_HIT, precision, max_steps: 255 /* :119 */,
            tile_w: 1, tile_h: 1, flags: STEPTH_FLAG_HEAVY_FIRST, ..Default::default() };
        let mut max = 0u32;
        let rc = ENGINE.with(|e| unsafe { stepth_depth_from_additional(e.raw(), &main, &addi, &params,
            groups.as_ptr(), groups.len(), STEPTH_MEM_HOST, &out, &mut max) });
        match rc {
            STEPTH_OK => {}
            STEPTH_ERR_SIZE_MISMATCH => return Err(io_invalid("Sizes don't match")),   // :45-48
            STEPTH_ERR_MAX_ZERO => panic!("attempt to divide by zero"),                // what :127 does today
            _ => return Err(io_invalid(&ENGINE.with(|e| e.last_error()))),
        }
        let depth_image = image::DynamicImage::ImageLuma8(
            image::ImageBuffer::from_raw(w, h, depth).unwrap())
            .resize(w, h, image::imageops::Gaussian)          // :133-134 stays on the CPU (or
            .to_luma8();                                      // STEPTH_FLAG_GAUSSIAN_TAIL: parity unpinned)
        self.load_depth(depth_image)                                             // :135
    }

    pub fn depth_split(&self, zones: u8) -> Vec<(Option<u8>, Option<u8>)> {      // :162-219
        let d = host_image(self.depth.as_raw().as_ptr(), self.width(), self.height(), 1);
        let mut out = [stepth_zone::default(); 256];
        let mut n = 0u32;
        let rc = ENGINE.with(|e| unsafe { stepth_depth_split(e.raw(), &d, zones as u32, out.as_mut_ptr(), 256, &mut n) });
        if rc != STEPTH_OK { panic!("depth_split: {}", ENGINE.with(|e| e.last_error())); } // -7/-8: where :203-217 panic today
        out[..n as usize].iter().map(|z| (if z.has_from != 0 { Some(z.from) } else { None },
                                          if z.has_to != 0 { Some(z.to) } else { None })).collect()
    }

    pub fn select_foreground(&mut self) -> MaskImage {                            // :221-224
        let (w, h) = (self.width(), self.height());
        let mut mask = vec![0u8; (w * h) as usize];
        let d = host_image(self.depth.as_raw().as_ptr(), w, h, 1);
        let m = host_image(mask.as_mut_ptr(), w, h, 1);
        let rc = ENGINE.with(|e| unsafe { stepth_select_foreground(e.raw(), &d, &m, std::ptr::null_mut()) });
        if rc != STEPTH_OK { panic!("select_foreground: {}", ENGINE.with(|e| e.last_error())); }
        MaskImage { image: self.image.clone(), mask: image::ImageBuffer::from_raw(w, h, mask).unwrap() }
    }

    pub fn invert_depth(&mut self) {                                              // :243-247
        let d = host_image(self.depth.as_raw().as_ptr(), self.width(), self.height(), 1);
        ENGINE.with(|e| unsafe { stepth_invert_depth(e.raw(), &d, 0) });
    }
}

// Many frames per call (video): stepth_depth_from_additional_batch / stepth_adaptive_groups_batch /
// stepth_depth_from_groups_batch take arrays of stepth_image (device memory for STEPTH_FLAG_ASYNC);
// stepth_select_foreground_batch does the same for the consumers.
*/

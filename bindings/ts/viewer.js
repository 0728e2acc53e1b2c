'use strict';
/*
 * viewer.js (+ viewer.d.ts) -- the reference's `Viewer` (src/viewer.ts:8-115) without the browser: forward pass in 'pointcloud' render
 * mode + rasterizer on setPointCloud (viewer.ts:46-66); per frame forward -> rasterize -> blitToTexture(swap-chain image) (72-87).
 * `canvas` is any object with `width` / `height` (an HTMLCanvasElement has them; `clientWidth` / `clientHeight` are honoured by resize
 * handling when present); `context` stands in for the GPUCanvasContext: an object whose getCurrentTexture() returns the rgba8 image to
 * present into ({ ptr, width, height }: a HipBuffer carrying its size), or null -- then the viewer owns its "swap-chain image"
 * (frameBuffer) and readFrame() / savePNG() take the place of presentation.  Camera interaction (CameraControl) is UI and stays out.
 *
 * A viewer and a trainer share one PointCloud in the reference (main.ts:389, 524).  While the trainer's optimizer defers its SH-DC writes
 * the cloud carries the compact array (pointCloud.dcWords) and this viewer's forward pass reads it by itself: the colours on screen are
 * the trained ones without any hand-over call.
 */
const fs = require('fs');
const hip = require('./webdgs_hip.js');
const { Camera } = require('./camera.js');
const { encodePNG } = require('./images.js');

class Viewer {
  constructor(device, context, canvas, format) {
    this.device = device; this.context = context || null; this.canvas = canvas; this.presentationFormat = format || 'rgba8unorm';
    this.forwardPass = null; this.rasterizer = null; this.pointCloud = null; this.frameBuffer = null;
    this.camera = new Camera(canvas, device);
    this.cameraControl = { update(_dt) {} };   // (camera-control.ts is UI: no interactive control here)
    this.handleResize();
  }
  setPointCloud(pointCloud) {   // viewer.ts:46-66
    this.settings = { renderMode: 'pointcloud' };   // (a new cloud starts in point-cloud mode, default scale and point size)
    this.tileEntries = 0;
    this.pointCloud = pointCloud;
    this.buildPasses();
    this.camera.on_update_canvas();
  }
  buildPasses() {
    if (this.forwardPass) this.forwardPass.destroy();
    if (this.rasterizer) this.rasterizer.destroy();
    this.forwardPass = new hip.TiledForwardPass(this.device, this.pointCloud, this.camera.uniform_buffer,
      { viewportWidth: this.canvas.width, viewportHeight: this.canvas.height, renderMode: 'pointcloud', maxTileEntries: this.tileEntries });
    this.rasterizer = new hip.TiledRasterizer({ device: this.device, forwardPass: this.forwardPass, format: this.presentationFormat });
    if (this.settings.renderMode !== undefined) this.forwardPass.setRenderMode(this.settings.renderMode);
    if (this.settings.gaussianScale !== undefined) this.forwardPass.setGaussianScale(this.settings.gaussianScale);
    if (this.settings.pointSize !== undefined) this.forwardPass.setPointSize(this.settings.pointSize);
  }
  update(dt) { this.cameraControl.update(dt); }
  /** The image this frame is presented into: the context's current texture, or the viewer's own frame buffer. */
  currentTexture() {
    if (this.context && this.context.getCurrentTexture) return this.context.getCurrentTexture();
    const w = this.canvas.width, h = this.canvas.height;
    if (!this.frameBuffer || this.frameBuffer.width !== w || this.frameBuffer.height !== h) {
      if (this.frameBuffer) this.frameBuffer.destroy();
      this.frameBuffer = this.device.createBuffer({ size: 4 * w * h, label: 'swap-chain image' });
      this.frameBuffer.width = w; this.frameBuffer.height = h;
    }
    return this.frameBuffer;
  }
  render(commandEncoder) {   // viewer.ts:72-87
    if (!this.forwardPass || !this.rasterizer || !this.pointCloud) return;
    this.forwardPass.encode(commandEncoder);
    const swap = this.currentTexture();
    this.rasterizer.encode(commandEncoder, swap.width, swap.height);
    this.rasterizer.blitToTexture(commandEncoder, swap);
  }
  setRenderMode(mode) { if (this.forwardPass) { this.settings.renderMode = mode; this.forwardPass.setRenderMode(mode); } }
  setGaussianScale(value) { if (this.forwardPass) { this.settings.gaussianScale = value; this.forwardPass.setGaussianScale(value); } }
  setPointSize(value) { if (this.forwardPass) { this.settings.pointSize = value; this.forwardPass.setPointSize(value); } }
  getForwardPass() { return this.forwardPass; }
  handleResize() {   // viewer.ts:106-113
    if (!this.canvas) return;
    if (this.canvas.clientWidth !== undefined) this.canvas.width = this.canvas.clientWidth;
    if (this.canvas.clientHeight !== undefined) this.canvas.height = this.canvas.clientHeight;
    this.camera.on_update_canvas();
    if (this.forwardPass) this.forwardPass.setViewport(this.canvas.width, this.canvas.height);
  }
  /** New canvas size (the ResizeObserver callback of viewer.ts:37-40). */
  resize(width, height) { this.canvas.width = width; this.canvas.height = height; if (this.canvas.clientWidth !== undefined) { this.canvas.clientWidth = width; this.canvas.clientHeight = height; } this.handleResize(); }
  /** The presented image as a Uint8Array of width * height * 4 bytes (synchronises). */
  readFrame() {
    // If the frame's tile-entry list outran what the library sized for the cloud (the reference would show the truncated picture; the library reports it),
    // the viewer's passes are rebuilt around larger lists and the frame is rendered again -- other owners' reports on the same device are left to them.
    for (let attempt = 0; attempt < 4 && this.forwardPass; attempt++) {   // this viewer's own pass: its word is consumed by its own check
      try {
        this.forwardPass.check();
        const left = this.device.capacityReports.take([this.forwardPass.handle]);   // (consumed by another owner's wait, left for us)
        if (left) throw left;
        break;
      } catch (e) {
        // (a report names up to four passes: this viewer's line)
        const own = BigInt(this.forwardPass.handle), re = /(\d+) entries needed, max_tile_entries = (\d+) \(forward pass (0x[0-9a-fA-F]+)\)/g, text = String(e && e.message);
        let m = null;
        if (e && e.code === 'WDGS_E_CAPACITY') for (let x = re.exec(text); x && !m; x = re.exec(text)) if (BigInt(x[3]) === own) m = x;
        if (!m) throw e;
        this.tileEntries = Math.min(Math.max(2 * Number(m[2]), Math.floor(Number(m[1]) * 1.5)), 0xFFFFF000);
        this.buildPasses();
        this.render(null);
      }
    }
    const f = this.currentTexture();
    return new Uint8Array(this.device.readBuffer(f, 4 * f.width * f.height));
  }
  savePNG(file) { const f = this.currentTexture(); fs.writeFileSync(file, encodePNG(this.readFrame(), f.width, f.height)); }
  destroy() {
    if (this.forwardPass) this.forwardPass.destroy();
    if (this.rasterizer) this.rasterizer.destroy();
    if (this.frameBuffer) this.frameBuffer.destroy();
    this.camera.destroy();
    this.forwardPass = this.rasterizer = this.frameBuffer = null;
  }
}

module.exports = { Viewer };

// Typings of jpeg.js: 8-bit Huffman JPEG (baseline / extended sequential / progressive) -> rgba8, the texels libjpeg-turbo produces.
export function decodeJPEG(bytes: Uint8Array | Buffer): { width: number; height: number; data: Uint8Array };

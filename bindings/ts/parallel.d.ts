// Typings of parallel.js: view-sharded data parallelism for the TypeScript-side host (SURVEY.md 8(e); no reference counterpart).
import { Communicator, HipBuffer, HipDevice } from './webdgs_hip';
export const GRAD_FLOATS: number;
export function slicePoints(numPoints: number, world: number): number;
export function ownedRange(numPoints: number, world: number, rank: number): { first: number; count: number };
export function shardViews(viewIds: number[], rank: number, world: number): number[];
export class Exchange {
  worldSize: number; rank: number; name: string; force: boolean;
  exchangeGradients(grad: HipBuffer, visible: HipBuffer, flag: HipBuffer | null, slicePoints: number): void;
  allgatherRows(rows: HipBuffer, slicePoints: number): void;
  broadcast(ptr: bigint, bytes: number, root: number): void;
  allreduceCounts(counts: HipBuffer, count: number): void;
  destroy(): void;
}
export class CapiExchange extends Exchange {
  constructor(device: HipDevice, uniqueId: ArrayBuffer, worldSize: number, rank: number);
  readonly comm: Communicator;
}
export function rendezvousId(file: string, rank: number, timeoutMs?: number): ArrayBuffer;
export function envRanks(): { rank: number; world: number; localRank: number };
export function defaultExchange(device: HipDevice): Exchange;

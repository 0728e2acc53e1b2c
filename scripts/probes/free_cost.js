const path='/root/repo/bindings/ts/webdgs_hip.js'; const hip=require(path);
const now=()=>{const t=process.hrtime();return (t[0]+t[1]*1e-9)*1e3;};
const dev=new hip.HipDevice(0);
function round(label){ const bs=[]; let t0=now(); for(let i=0;i<50;i++) bs.push(dev.createBuffer({size:1<<20})); dev.synchronize(); const c=(now()-t0)/50; t0=now(); for(const b of bs) b.destroy(); const d=(now()-t0)/50; console.log(label,'create ms',c.toFixed(4),'destroy ms',d.toFixed(4)); }
round('node fresh'); round('node again');
(async()=>{ await dev.queue.onSubmittedWorkDone(); round('node after a Promise-based wait'); round('node again'); dev.destroy(); })();

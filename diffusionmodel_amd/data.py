"""Attention-mask generation of the reference's CrackDataset on the device (new_scripy.py:533-546).

The VOC-XML / image I/O of the dataset stays out of scope (host I/O); what the loss consumes —
the (B,S,S) mask with values LOW / MID (lower half) / HIGH (inside the scaled bounding box) — is rasterised
by one kernel from the boxes, so a loader only has to ship 4 integers per sample.
"""
import torch

from ._lib import call, ptr
from .config import Cfg


def scaled_bbox(xmin, ymin, xmax, ymax, orig_w, orig_h, size=None):
    """Scale + clamp exactly like new_scripy.py:541-544 (Python round(), i.e. round-half-to-even)."""
    size = Cfg.IMG_SIZE if size is None else size

    def cl(v):
        return max(0, min(size - 1, v))
    return (cl(round(xmin * size / orig_w)), cl(round(ymin * size / orig_h)),
            cl(round(xmax * size / orig_w)), cl(round(ymax * size / orig_h)))


def attn_masks(boxes_scaled, size=None, device="cuda:0"):
    """boxes_scaled: sequence of (x0, y0, x1, y1) from `scaled_bbox` -> (B, S, S) fp32 mask on `device`."""
    size = Cfg.IMG_SIZE if size is None else size
    boxes = torch.tensor(list(boxes_scaled), dtype=torch.int32).reshape(-1, 4).to(device)
    out = torch.empty((boxes.shape[0], size, size), dtype=torch.float32, device=device)
    call("dm_attn_mask", ptr(boxes), ptr(out), boxes.shape[0], size, float(Cfg.LOW_WEIGHT), float(Cfg.MID_WEIGHT), float(Cfg.HIGH_WEIGHT))
    return out


class CrackDataset(torch.utils.data.Dataset):
    """The reference's dataset (new_scripy.py:479-551) without torchvision: `root/images/<class>/*.{png,jpg,jpeg}` with one
    VOC-XML per image in `root/annotations/`; item = (image (3,S,S) fp32 normalised with NORM_MEAN/STD, label, attn_mask (S,S)).

    Resize = PIL bilinear to (S, S) followed by /255 and (x - mean) / std, i.e. transforms.Resize((S,S)) + ToTensor + Normalize.
    `return_boxes=True` returns the scaled box (x0, y0, x1, y1) instead of the rasterised mask, so that a training loop can
    build the masks of a whole batch on the device with `attn_masks` (one kernel instead of S*S host writes per sample)."""

    EXT = (".png", ".jpg", ".jpeg")

    def __init__(self, root_dir, img_size=None, return_boxes=False):
        import os
        self.root_dir, self.size, self.return_boxes = root_dir, (Cfg.IMG_SIZE if img_size is None else img_size), return_boxes
        img_root = os.path.join(root_dir, "images")
        self.classes = sorted(d for d in os.listdir(img_root) if os.path.isdir(os.path.join(img_root, d)))
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for cname in self.classes:
            cdir = os.path.join(img_root, cname)
            for name in sorted(os.listdir(cdir)):
                if name.endswith(self.EXT):
                    xml = os.path.join(root_dir, "annotations", name.rsplit(".", 1)[0] + ".xml")
                    if os.path.exists(xml):
                        self.samples.append((os.path.join(cdir, name), xml, self.class_to_idx[cname]))

    def __len__(self):
        return len(self.samples)

    @staticmethod
    def read_box(xml_path):
        """(xmin, ymin, xmax, ymax, width, height) of the first <bndbox>, as the reference reads them."""
        import xml.etree.ElementTree as ET
        root = ET.parse(xml_path).getroot()
        bb = root.find(".//bndbox")
        vals = [int(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]
        return (*vals, int(root.find(".//width").text), int(root.find(".//height").text))

    def __getitem__(self, idx):
        import numpy as np
        from PIL import Image
        img_path, xml_path, label = self.samples[idx]
        S = self.size
        im = Image.open(img_path).convert("RGB").resize((S, S), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()
        mean = torch.tensor(Cfg.NORM_MEAN).view(3, 1, 1)
        std = torch.tensor(Cfg.NORM_STD).view(3, 1, 1)
        x = (x - mean) / std
        box = scaled_bbox(*self.read_box(xml_path), size=S)
        if self.return_boxes:
            return x, label, torch.tensor(box, dtype=torch.int32)
        mask = torch.full((S, S), float(Cfg.LOW_WEIGHT))
        mask[S // 2:, :] = float(Cfg.MID_WEIGHT)
        x0, y0, x1, y1 = box
        mask[y0:y1, x0:x1] = float(Cfg.HIGH_WEIGHT)
        return x, label, mask


def convert_supervisely(src_split_dir, dst_root, link=True):
    """Lay a DatasetNinja / Supervisely split (`<split>/img/*.jpg` + `<split>/ann/*.jpg.json`, the format of the road-damage
    dataset bundled with the reference) out the way CrackDataset expects: `images/<class>/<name>` + `annotations/<stem>.xml`.
    Class and box come from the first rectangle object of the annotation (the reference reads one box per image).
    Returns the number of images written."""
    import json
    import os
    import shutil
    n = 0
    ann_dir, img_dir = os.path.join(src_split_dir, "ann"), os.path.join(src_split_dir, "img")
    os.makedirs(os.path.join(dst_root, "annotations"), exist_ok=True)
    for fn in sorted(os.listdir(ann_dir)):
        if not fn.endswith(".json"):
            continue
        ann = json.load(open(os.path.join(ann_dir, fn)))
        rects = [o for o in ann.get("objects", []) if o.get("geometryType") == "rectangle"]
        img_name = fn[:-5]
        src_img = os.path.join(img_dir, img_name)
        if not rects or not os.path.exists(src_img):
            continue
        obj = rects[0]
        (xa, ya), (xb, yb) = obj["points"]["exterior"][:2]
        cls = obj["classTitle"].replace(" ", "_")
        cdir = os.path.join(dst_root, "images", cls)
        os.makedirs(cdir, exist_ok=True)
        dst_img = os.path.join(cdir, img_name)
        if not os.path.exists(dst_img):
            if link:
                os.symlink(os.path.abspath(src_img), dst_img)
            else:
                shutil.copyfile(src_img, dst_img)
        w, h = ann["size"]["width"], ann["size"]["height"]
        xml = (f"<annotation><filename>{img_name}</filename><size><width>{w}</width><height>{h}</height><depth>3</depth></size>"
               f"<object><name>{cls}</name><bndbox><xmin>{min(xa, xb)}</xmin><ymin>{min(ya, yb)}</ymin>"
               f"<xmax>{max(xa, xb)}</xmax><ymax>{max(ya, yb)}</ymax></bndbox></object></annotation>")
        with open(os.path.join(dst_root, "annotations", img_name.rsplit(".", 1)[0] + ".xml"), "w") as f:
            f.write(xml)
        n += 1
    return n

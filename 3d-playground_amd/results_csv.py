"""The two on-disk wire formats of the callers of the detector path, as drop-ins:

  write_detections_csv(data_list, sequence, fps)   <- perform_3D_detection_on_video_sequences.py:142-307
  write_results_csv(self)                          <- MC_Crop_Tracker.write_results_csv, MC3D_crop_tracker.py:1318-1453

Both are host-side row formatting (the reference uses ``csv.writer``, so a cell is ``str()`` of whatever Python / numpy
scalar the code appended: float32 cells print the shortest float32 representation, float64 cells the shortest double
one).  What changes is where the numbers come from: the reference converts every track state one at a time
(``hg.state_to_space`` and ``hg.state_to_im`` per row, :1409-1414 -- two tiny tensor programs per row, 7 087 rows in the
result files it ships); here the qualifying rows are gathered first, ONE batched ``state_to_space`` and ONE batched
``state_to_im`` run on the device (``homography.hip``; per-object arithmetic, so the values are those of the row-wise
calls) and the formatting loop only reads arrays.  ``append_detections`` is the per-frame half of the detection file:
one device->host copy of the survivors instead of three.

The row builders (``results_rows`` / ``detection_rows``) are separate from the file writers so that tests can compare
rows with the reference's shipped ``3D_tracking_results*.csv`` / ``working_3D_tracking_data.csv`` cell by cell.
"""
import csv
import os

import numpy as np
import torch

CLASS_NAMES = {0: "sedan", 1: "midsize", 2: "van", 3: "pickup", 4: "semi", 5: "truck (other)", 6: "motorcycle",
               7: "trailer"}                                  # perform_3D_detection_on_video_sequences.py:157-164

DETECTIONS_HEADER = [                                         # perform_3D_detection_on_video_sequences.py:209-240
    "Frame #", "Timestamp", "Object Confidence", "Object class", "BBox xmin", "BBox ymin", "BBox xmax", "BBox ymax",
    "vel_x", "vel_y", "Generation method", "GPS lat of bbox bottom center", "GPS long of bbox bottom center",
    "fbrx", "fbry", "fblx", "fbly", "bbrx", "bbry", "bblx", "bbly", "ftrx", "ftry", "ftlx", "ftly", "btrx", "btry",
    "btlx", "btly"]

RESULTS_HEADER = [                                            # MC3D_crop_tracker.py:1333-1379 (+ the ts_bias column, :1380)
    "Frame #", "Timestamp", "Object ID", "Object class", "BBox xmin", "BBox ymin", "BBox xmax", "BBox ymax", "vel_x",
    "vel_y", "Generation method", "fbrx", "fbry", "fblx", "fbly", "bbrx", "bbry", "bblx", "bbly", "ftrx", "ftry", "ftlx",
    "ftly", "btrx", "btry", "btlx", "btly", "fbr_x", "fbr_y", "fbl_x", "fbl_y", "bbr_x", "bbr_y", "bbl_x", "bbl_y",
    "direction", "camera", "acceleration", "speed", "veh rear x", "veh center y", "theta", "width", "length", "height"]


# ------------------------------------------------------------------------------------------------ detections file
def append_detections(data_list, frame_idx, timestamp, sequence, boxes, scores, labels):
    """The per-frame tail of detect_video_sequence (perform_3D_detection_on_video_sequences.py:128-133): one datum
    [frame_idx, timestamp, sequence, box[20] f32, score f32, label i64] per surviving detection.  boxes / scores / labels
    may be device tensors: they leave the device in ONE copy."""
    n = int(boxes.shape[0])
    if n == 0:
        return data_list
    packed = torch.cat((boxes.reshape(n, -1).float(), scores.reshape(n, 1).float(), labels.reshape(n, 1).float()), dim=1)
    host = packed.detach().cpu().numpy()
    b, s, l = host[:, :-2], host[:, -2], host[:, -1].astype(np.int64)       # class ids < 2^24: exact through float32
    for j in range(n):
        data_list.append([frame_idx, timestamp, sequence, b[j], s[j], l[j]])
    return data_list


def detection_rows(data_list):
    """Main-chunk rows of write_detections_csv (:262-304), one list of cells per datum."""
    rows = []
    for frame_idx, timestamp, _, bbox, conf, class_idx in data_list:
        cls = class_idx.item() if hasattr(class_idx, "item") else class_idx
        row = [frame_idx, timestamp, conf, CLASS_NAMES[cls], bbox[16], bbox[17], bbox[18], bbox[19], "---", "---",
               "3D Detector", "---", "---"]
        for k in (2, 3, 0, 1, 6, 7, 4, 5, 10, 11, 8, 9, 14, 15, 12, 13):      # fbr, fbl, bbr, bbl, ftr, ftl, btr, btl
            row.append(bbox[k])
        rows.append(row)
    return rows


def write_detections_csv(data_list, sequence, fps, out_dir="_outputs"):
    """perform_3D_detection_on_video_sequences.py:142-307: summary, fps and parameter chunks, then one row per detection."""
    outfile = os.path.join(out_dir, sequence.split("/")[-1].split(".")[0] + "_3D_detections.csv")
    with open(outfile, mode="w") as f:
        out = csv.writer(f, delimiter=",")
        out.writerow(["Video sequence name", "Processing start time", "Processing end time", "Timestamp start time",
                      "Timestamp end time", "Unique objects", "GPU"])
        out.writerow([sequence, "---", "---", data_list[0][1], data_list[-1][1], "---", "---"])
        out.writerow([])
        out.writerow(["Processing fps"])
        out.writerow([fps])
        out.writerow([])
        out.writerow(["Confidence Cutoff", "NMS Cutoff"])
        out.writerow([0.3, 0.5])
        out.writerow([])
        out.writerow(DETECTIONS_HEADER)
        out.writerows(detection_rows(data_list))
    return outfile


# ------------------------------------------------------------------------------------------------ tracking results file
def results_rows(ids, timestamps, states, space, im, class_names, ts_bias, camera="p1c1", gen="3D Detector"):
    """Rows of write_results_csv (:1395-1453) from arrays: states [n,7] float32 (x, y, l, w, h, direction, speed),
    space [n,4,2] float32 (the first four road-plane corners), im [n,8,2] float64 (image corners)."""
    rows = []
    for i in range(len(ids)):
        st, b3 = states[i], im[i]
        row = ["-", timestamps[i], ids[i], class_names[i],
               b3[:, 0].min().item(), b3[:, 1].min().item(), b3[:, 0].max().item(), b3[:, 1].max().item(), 0, 0, gen]
        row += list(b3.reshape(-1)) + list(space[i].reshape(-1))
        row += [st[5], camera, 0, st[6], st[0], st[1], np.pi / 2.0 if st[5] == -1 else 0, st[3], st[2], st[4], ts_bias[i]]
        rows.append(row)
    return rows


def select_tracks(self):
    """The rows write_results_csv keeps (:1400-1404): tracks longer than f_init with a non-zero x -> (indices, states)."""
    keep, states = [], []
    for i, item in enumerate(self.all_tracks):
        if len(self.all_classes[item[0]]) > self.f_init:
            st = torch.as_tensor(item[2]).float()
            if st[0] != 0:
                keep.append(i)
                states.append(st)
    return keep, states


def write_results_csv(self):
    """Drop-in for MC_Crop_Tracker.write_results_csv (MC3D_crop_tracker.py:1318-1453).  Reads self.all_tracks,
    self.all_classes, self.f_init, self.class_dict, self.all_ts_bias, self.cameras, self.hg, self.output_file."""
    camera = "p1c1"                                           # "default dummy value" (:1392)
    keep, states = select_tracks(self)
    header = RESULTS_HEADER + ["ts_bias for cameras {}".format(self.cameras)]
    with open(self.output_file, mode="w") as f:
        out = csv.writer(f, delimiter=",")
        out.writerow(header)
        if not keep:
            return
        st = torch.stack(states)                                                    # [n,7] float32
        space = self.hg.state_to_space(st)[:, :4, :2].cpu().numpy()                 # one launch for every row (:1409)
        im = self.hg.state_to_im(st, name=camera).cpu().numpy()                     # one launch for every row (:1414)
        ids = [self.all_tracks[i][0] for i in keep]
        names = [self.class_dict[np.argmax(self.all_classes[k])] for k in ids]
        out.writerows(results_rows(ids, [self.all_tracks[i][1] for i in keep], st.numpy(), space, im, names,
                                   [self.all_ts_bias[i] for i in keep], camera=camera))

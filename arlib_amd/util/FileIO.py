"""Text I/O with the reference's interface (util/FileIO.py:23-32): `user item rating` per line."""
import os


class FileIO(object):
    @staticmethod
    def load_data_set(file):
        data = []
        with open(file) as f:
            for line in f:
                parts = line.strip().split(' ')
                data.append([parts[0], parts[1], float(parts[2])])
        return data

    @staticmethod
    def write_file(dir, file, content, op='w'):
        os.makedirs(dir, exist_ok=True)
        with open(dir + file, op) as f:
            f.writelines(content)

    @staticmethod
    def delete_file(file_path):
        if os.path.exists(file_path):
            os.remove(file_path)
